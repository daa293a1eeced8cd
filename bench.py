#!/usr/bin/env python3
"""bench.py -- rays/sec and PSFs/sec of the dual-pixel ray-traced PSF path.

Workload (BASELINE.json configs[1]): rf50mm refocused to 1 m (F/4 stop),
32x32x16 (x, y, z) PSF volume = 16384 point sources per GPU, 4096 primary rays
per point, 65x65 LEFT and RIGHT PSFs, lambda = 0.589 um.  One "step" = one
Lensgroup.psf_lr call over the rank's 16384 points: draw the pupil uniforms
(torch CPU generator, the reference's order), chief-ray centre pass (2048 rays
per point) and the sample->trace->splat->normalise pass in ONE fused kernel launch,
verification of the batch-global Newton trip counts (of step i while step i+1 runs).  For N > 1
the ranks share the pupil sample set (48 KB broadcast) and the trip check (mask all-reduce), and
the PSF shards are all-gathered to every rank over RCCL (north_star's reassembly step; on a side
stream under the next step's kernels) -- `value` includes it, `value_no_gather` is the rate of the
same loop without it; `gather` carries the bytes a rank receives per step, the all-gather's own time on its stream
and `gather_bound` (does it take longer than the kernel it hides under?).  Weak scaling (default): every rank
renders its own 16384-point slab of a 32x32x(16*N) volume; `--scaling strong`: the ONE 16384-point volume (c3:
65536 points) cut into N contiguous shards.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling weak|strong]   (N > 1: starts its own N ranks)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (see the driver contract); `roofline` describes the
dominant kernel (k_psf_lr), `cpu_baseline` times oracle/ (a C port of the
reference's CPU path, OpenMP over the host cores, and a PyTorch-CPU restatement of
the reference's whole-tensor execution model) on a bounded sample of the
same workload.  oracle/ is only loaded for that baseline leg.
"""
import argparse
import json
import math
import os
import sys
import time

# libgomp reads this when it initialises: idle OpenMP threads of the CPU-baseline
# leg must sleep, not spin, inside a CPU-quota'd container
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
# the pool's host driver only supports dmabuf IPC (RCCL between processes needs it)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# HIP maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and packets of one hardware queue
# run in order: a rank of a multi-GPU run has eight streams (render, pupil, read-back, gather, three RCCL
# communicators, sampling), and on four queues the next step's kernel sat behind the previous step's mask all-reduce
# although their streams are independent -- 0.09 ms of idle GPU per step, 4 us with 16 queues (`--workload sweep`,
# profiles/r05/sweep_hw_queues.txt).  Read when the HIP runtime initialises, i.e. before anything touches the GPU.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

KS, SPP, GRID_XY, GRID_Z = 65, 4096, 32, 16
DP = (0.78, 1.44, 0.3, 0.5)
# BASELINE.json configs: c2 is the headline (and the default); c3 / c4 are the other GPU
# configurations, runnable with --workload for completeness (they are parity-test cases).
WORKLOADS = {
    "c2": dict(lens="rf50mm", ks=65, spp=4096, grid_z=16, sensor_z=62.25,
               desc="rf50mm 32x32x{gz} (x,y,z) PSF volume"),
    "c3": dict(lens="rf50mm", ks=21, spp=8192, grid_z=64, sensor_z=62.25,
               desc="rf50mm dense PSFNet grid 32x32x{gz}, Gaussian-warped depth planes around the 1 m "
                    "focal plane (psfnet.py:229-232) (8 shards of 8192 points on a node)"),
    "c3k65": dict(lens="rf50mm", ks=65, spp=8192, grid_z=64, sensor_z=62.25,
                  desc="rf50mm dense PSFNet grid 32x32x{gz} as c3, 65x65 grids (config 3's second kernel size)"),
    "c4": dict(lens="rf35mm", ks=65, spp=4096, grid_z=16, sensor_z=80.447,
               desc="rf35mm (21 surfaces) 32x32x{gz} PSF volume"),
}
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s HBM3E
VALU_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: vector fp32 (FMA counted as 2; the parity contract forbids contraction)
# the two workloads either side of the hot path: not PSF-volume renders
EXTRA_WORKLOADS = ("f1", "tcp", "staged", "sweep", "c5")


_JSON_FD = None


def claim_stdout():
    """The contract is ONE JSON line on stdout.  Native libraries print there too (RCCL greets with a five-line version
    banner when its first communicator comes up; MIOpen logs): from here on file descriptor 1 leads to stderr, and the
    JSON line goes to the descriptor stdout had when the process started."""
    global _JSON_FD
    if _JSON_FD is None:
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)


def emit_line(res):
    line = (json.dumps(res) + "\n").encode()
    if _JSON_FD is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_JSON_FD, line)


def source_hash():
    """Hash of what the loaded library was built from (csrc/*.hip, *.hpp, the Makefile with its flags, the ABI header):
    carried PMC counters name the hash they were collected on; a mismatch is reported as `stale`."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "sdirt_amd", "csrc", "*.hip")) +
                   glob.glob(os.path.join(ROOT, "sdirt_amd", "csrc", "*.hpp")) +
                   [os.path.join(ROOT, "include", "sdirt_dp.h"), os.path.join(ROOT, "sdirt_amd", "csrc", "Makefile")])
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


FOC_Z_RF50 = (-1000.0 + 62.25 - (-200.0)) / (-20000.0 - (-200.0))   # psfnet.py:50-51, foc_z_arr[1]


def volume_points(world, workload="c2"):
    """The reference's test grid (psfnet.py:220-226) x depths z2depth(z) (psfnet.py:36-37,
    724-726): [32*32*GRID_Z*world, 3], z-major so that a rank's contiguous shard is a slab of
    depth planes.  c2 / c4: z = linspace(0, 1).  c3 (the PSFNet evaluation grid): the
    reference's Gaussian-warped spacing around the focal plane (psfnet.py:229-232,
    foc_z = foc_z_arr[1]) -- planes crowd around 1 m, where the PSF changes fastest."""
    g = GRID_XY
    dense = workload.startswith("c3")
    nz = (8 if dense else WORKLOADS[workload]["grid_z"]) * world
    x, y = torch.meshgrid(torch.linspace(-1 + 1 / (2 * g), 1 - 1 / (2 * g), g),
                          torch.linspace(1 - 1 / (2 * g), -1 + 1 / (2 * g), g), indexing="xy")
    if dense:
        z_gauss = torch.linspace(-3, 3, nz)
        z = torch.zeros_like(z_gauss)
        z[z_gauss > 0] = (1 - FOC_Z_RF50) * z_gauss[z_gauss > 0] / 3 + FOC_Z_RF50
        z[z_gauss < 0] = FOC_Z_RF50 * z_gauss[z_gauss < 0] / 3 + FOC_Z_RF50
    else:
        z = torch.linspace(0, 1, nz)
    depth = z * (-20000.0 - (-200.0)) + (-200.0)
    pts = torch.stack([x.reshape(1, -1).expand(len(z), -1), y.reshape(1, -1).expand(len(z), -1),
                       depth.reshape(-1, 1).expand(-1, g * g)], dim=-1)
    return pts.reshape(-1, 3).contiguous()


def build_lens(device, name="rf50mm", sensor_z=62.25):
    """The lens as 1_fit_psfnet.py:21-25 sets it up: sensor at 62.25 mm (rf50mm) /
    80.447 mm (rf35mm) (psfnet.py:42-45), then refocus to 1 m -- all with this
    package's own geometric optics running on the GPU."""
    from sdirt_amd import Lensgroup
    lens = Lensgroup(os.path.join(ROOT, "sdirt_amd", "data", f"{name}.json"),
                     sensor_res=(512, 768), post_computation=False, device=device)
    lens.d_sensor = sensor_z
    torch.manual_seed(0)
    lens.post_computation()
    lens.refocus(-1000 + lens.d_sensor)
    return lens


def available_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, math.ceil(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def lens_state(lens):
    """The scalars and the surface table of `lens` in the form the oracle takes."""
    st = dict(hfov=lens.hfov, r_last=float(lens.r_last), sensor_size=list(lens.sensor_size),
              pupil_z=lens.entrance_pupil()[0], pupil_r=lens.entrance_pupil()[1],
              d_sensor=lens.d_sensor, pixel_size=lens.pixel_size, surfaces=[])
    for s in lens.surfaces:
        st["surfaces"].append(dict(kind={0: "plane", 1: "sphere", 2: "asphere"}[s.kind], r=s.r,
                                   d=float(s.d), c=float(s.c), k=float(s.k),
                                   ai=[float(a) for a in (s.ai if s.ai is not None else [])],
                                   n1={repr(0.589): s.mat1.ior(0.589)},
                                   n2={repr(0.589): s.mat2.ior(0.589)}))
    return st


def cpu_baseline(lens, points, pupil, budget_s=12.0):
    """Times the two CPU restatements of the reference path on every k-th point of the same
    volume with the pupil sample points of the GPU leg's LAST step (`pupil` = x2, y2, xc, yc as the
    device mapped them), on the cores this process may use:
      * `value` (kind "port"): oracle/sdirt_oracle.c, a per-ray C port with OpenMP -- the
        strongest CPU implementation of the path we have;
      * `torch`: oracle/torch_port.py, the reference's own execution model (whole-tensor fp32
        PyTorch ops, 256 points per pass, torch.set_num_threads(cores))."""
    from oracle import oracle as orc
    from oracle import torch_port as tp
    st = lens_state(lens)
    cores = available_cores()
    orc.set_num_threads(cores)
    x2, y2, xc, yc = [np.ascontiguousarray(v.detach().cpu().numpy(), dtype=np.float32) for v in pupil]
    assert len(x2) == SPP and len(xc) == 2048
    pts = points.numpy()

    def subset(n):
        return pts[:: max(1, len(pts) // n)][:n]

    def run(n):
        sel = subset(n)
        t0 = time.perf_counter()
        orc.psf(st, sel, x2, y2, xc, yc, KS, dp=list(DP))
        return len(sel), time.perf_counter() - t0
    n0, t0 = run(64)                                    # calibration (also warms the threads)
    n = int(min(len(pts), max(64, n0 * budget_s / max(t0, 1e-3))))
    n = min(n, 16384)                                   # bound memory: [S, n, 3] fp32 x 2 = 2.1 GB
    n1, t1 = run(n)
    res = {"value": n1 * SPP / t1, "unit": "rays/s", "cores": cores, "kind": "port",
           "cpu_model": cpu_model(), "psfs_per_s": n1 / t1,
           "sample": f"{n1} of the {len(pts)} point sources (every {max(1, len(pts) // n)}-th), "
                     f"{SPP} spp + 2048 chief-ray rays each, ks {KS}, L+R, {t1:.1f} s; "
                     "oracle/sdirt_oracle.c with OpenMP"}

    old_threads = torch.get_num_threads()
    torch.set_num_threads(cores)
    try:
        def run_t(n):
            sel = subset(n)
            t0 = time.perf_counter()
            tp.psf(st, sel, x2, y2, xc, yc, KS, dp=list(DP), chunk=256)
            return len(sel), time.perf_counter() - t0
        run_t(64)                                       # thread pool / allocator warm-up
        m0, u0 = run_t(256)
        m = int(min(len(pts), max(256, (m0 * budget_s / max(u0, 1e-3)) // 256 * 256)))
        m1, u1 = run_t(m) if m > m0 else (m0, u0)
        res["torch"] = {"value": m1 * SPP / u1, "unit": "rays/s", "threads": cores,
                        "psfs_per_s": m1 / u1, "torch": torch.__version__,
                        "sample": f"{m1} point sources in passes of 256, {u1:.1f} s; "
                                  "oracle/torch_port.py (builder's PyTorch-CPU restatement of the "
                                  "reference's whole-tensor op sequence)"}
    finally:
        torch.set_num_threads(old_threads)
    return res


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) without a launcher: start N fresh rank processes with
    torch.distributed.run and relay their output and exit code.  Runs BEFORE anything here has
    touched the GPU (this process never does: it only counts devices)."""
    import socket
    import subprocess
    backend = os.environ.get("SDIRT_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()            # does not initialise the GPU
    if backend == "nccl" and ndev < args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but {ndev} GPU(s) visible.  (Dry run of the "
                         "multi-rank control flow on fewer GPUs: SDIRT_BENCH_BACKEND=gloo; its "
                         "numbers mean nothing.)\n")
        return 2
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=dict(os.environ))


def pmc_counters(workload):
    """PMC counters of k_psf_lr carried over from a rocprofv3 --pmc run of this same command
    (counters cannot be read from inside the process): newest profiles/rNN/pmc_<workload>.json."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", f"pmc_{workload}.json")))
    if files:
        with open(files[-1]) as f:
            c = json.load(f)
        c["file"] = os.path.relpath(files[-1], ROOT)
        c["stale"] = c.get("source_hash") != source_hash()
        return c
    return None


def _issue_bound(n_valu, n_trans, simd_cycles):
    if not n_trans:
        return None
    need = (n_valu - n_trans) * 2.25 + n_trans * 8.2
    return {"plain_cycles": 2.25, "transcendental_cycles": 8.2, "transcendental_instructions_per_launch": n_trans,
            "issue_cycles_needed": need, "simd_cycles_available": simd_cycles, "frac": need / simd_cycles}


def _hip_ms(fn, steps, warmup, device):
    """steps calls of fn between two synchronisations -> (wall ms per call, mean HIP-event ms per call)."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize(device)
    import gc
    gc.collect()
    gc.freeze()           # no full collection of torch's heap (40-55 ms) inside the timed calls, see main()
    ev = []
    t0 = time.perf_counter()
    for _ in range(steps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream(device))
        fn()
        e1.record(torch.cuda.current_stream(device))
        ev.append((e0, e1))
    torch.cuda.synchronize(device)
    wall = (time.perf_counter() - t0) / steps * 1e3
    return wall, float(np.mean([a.elapsed_time(b) for a, b in ev]))


def bench_f1(args, emit=True):
    """local_psf_render_fast (render_psf.py:120-155) -- per-pixel left/right PSF
    convolution of one 512 x 768 RGB frame with ks 21 kernels, the image-simulation step of
    2_dfdp_net.py.  HBM-bound: every pixel's [2, 21, 21] fp32 kernels are read exactly once."""
    from sdirt_amd.render_psf import local_psf_render_fast
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback exists)"
    dev = torch.device("cuda", 0)
    H, W, ks = 512, 768, 21
    g = torch.Generator(device=dev).manual_seed(0)
    img = torch.rand(1, 3, H, W, device=dev, generator=g)
    psf = torch.rand(1, H, W, 2, ks, ks, device=dev, generator=g)
    psf = psf / psf.sum((-1, -2), keepdim=True)
    fn = lambda: local_psf_render_fast(img, psf, ks)
    wall, kern = _hip_ms(fn, args.steps, max(args.warmup, 3), dev)
    k_sus = max(args.steps, int(args.sustain_seconds / max(wall * 1e-3, 1e-6))) if args.sustain_seconds > 0 else 0
    sus = _hip_ms(fn, k_sus, 0, dev)[0] if k_sus else None
    alg_bytes = psf.numel() * 4 + img.numel() * 4 + 2 * img.numel() * 4
    ach = alg_bytes / (kern * 1e-3) / 1e9
    res = {"metric": "pixels/sec per-pixel DP-PSF convolution 512x768 RGB ks21 (local_psf_render_fast)",
           "value": H * W / (wall * 1e-3), "unit": "pixels/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": wall, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16",
           "data": "synthetic",
           "config": {"workload": "one 512x768 RGB frame, per-pixel L/R kernels [1,512,768,2,21,21] fp32 (1.39 GB), "
                                  "fp16 arithmetic of the _fast renderer, replicate padding", "name": "f1"},
           "kernels_ms": {"local_psf_render": kern},
           "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                        "traffic": None, "kernel": "k_local_psf_render_wave<3, half, 21, 64>",
                        "algorithmic_bytes_per_launch": alg_bytes}}
    if sus is not None:
        res["ms_per_step_sustained"] = sus
    if emit:
        emit_line(res)
    return res


# 4096 points x 4096 spp x 28 B = 470 MB of rays (32 B = 537 MB with the obliquity array): well beyond the 256 MiB Infinity Cache, so that the streaming kernels
# of the chain are timed against HBM and not against the cache
STAGED_N, STAGED_SPP = 4096, 4096
KERNEL_OF = {"sample_rays": "k_sample_rays", "trace": "k_trace", "propagate_to": "k_propagate",
             "forward_integral": "k_forward_integral", "psf_normalize": "k_psf_normalize", "chief_center": "k_chief_center"}


def bench_staged(args, emit=True, lens=None, ks_list=None):
    """The API-compatible STAGED sequence of the reference (optics.py:460-494 sample_from_points, :889-904 psf_center,
    :638-664 trace2sensor, monte_carlo.py:9-68 forward_integral, optics.py:983-987 normalise) as the library calls a
    caller of those functions makes, rays held in HBM as a point-major SoA bundle (7 arrays of 4 bytes per ray: o, d, ra; the obliquity
    array, which nothing on this path reads, is not carried):
        sdirt_sample_rays -> sdirt_chief_center -> sdirt_trace -> sdirt_propagate_to -> sdirt_forward_integral ->
        sdirt_psf_normalize (L, R)
    on 4096 points of the config-2 volume (every 4th: all 16 depth planes) x 4096 spp, for 65x65 and 21x21 grids.
    One "step" = the whole chain once; every call is bracketed by HIP events on the stream it is launched on.  The same
    PSFs through the two fused entries (sdirt_trace2sensor, SDIRT_PSF_NORMALIZE on sdirt_forward_integral) are timed beside
    it (`fused_calls`).
    Algorithmic bytes per ray: sampler write o, d, ra = 28 B; trace read 28 + write 28; propagate read
    o, d_xyz = 24 + write o = 12; forward_integral read ox, oy, dx, dz, ra = 20 + the grids written once."""
    import ctypes as C
    from sdirt_amd import _lib
    from sdirt_amd.basics import Ray, dptr, stream_ptr
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback exists)"
    dev = torch.device("cuda", torch.cuda.current_device())
    if lens is None:
        lens = build_lens(dev)
    if ks_list is None:
        ks_list = tuple(int(v) for v in getattr(args, "staged_ks", "65,21").split(","))
    h, st = _lib.lib(), stream_ptr(dev)
    N, S = STAGED_N, STAGED_SPP
    M = N * S
    K = len(lens.surfaces)
    pts_all = volume_points(1, "c2")
    pts = pts_all[:: len(pts_all) // N][:N].contiguous()
    po = lens._points_to_object(pts)
    torch.manual_seed(0)
    # discovery through the package's own calls: the verified Newton trip tables become lens state
    ray = Ray.empty((S, N), 0.589, dev)
    pupilz, pupilr = lens.entrance_pupil()
    x2, y2 = lens._pupil_samples(S, pupilr)
    pupilz_c, pupilr_c = lens.entrance_pupil(shrink_pupil=True)
    xc, yc = lens._pupil_samples(2048, pupilr_c)
    cen = torch.empty((N, 2), dtype=torch.float32, device=dev)
    lens._chief_center(po, xc, yc, pupilz_c, cen)
    _lib.check(h.sdirt_sample_rays(dptr(po), N, dptr(x2), dptr(y2), S, float(pupilz), ray.c_rays(), st))
    lens.trace(ray, forward=True)
    t_trace = lens.trips.cache[("trace", round(float(ray.wvln), 6), 0, K, True, lens.precision)]
    t_cen = lens.trips.cache[("center", lens.precision)]
    trips_t = (C.c_int32 * K)(*[int(v) for v in t_trace])
    trips_c = (C.c_int32 * K)(*[int(v) for v in t_cen])
    handle = lens.dev_lens(0.589)
    mask_t, mask_c = lens._mask_buffer(), lens._mask_buffer()
    anyv = torch.zeros(1, dtype=torch.int32, device=dev)
    dp = _lib.DpParams(*DP)
    flags = lens._math_flags()
    stream = torch.cuda.current_stream(dev)
    out = {}
    for ks in ks_list:
        # L and R as the two halves of one buffer: a caller who owns both normalises them with ONE sdirt_psf_normalize
        # launch over 2 N tiles (each tile by its own maximum, optics.py:983-987)
        LR = torch.empty((2, N, ks, ks), dtype=torch.float32, device=dev)
        L, R = LR[0], LR[1]
        calls = [
            ("sample_rays", lambda: h.sdirt_sample_rays(dptr(po), N, dptr(x2), dptr(y2), S, float(pupilz), ray.c_rays(), st)),
            ("chief_center", lambda: h.sdirt_chief_center(handle, dptr(po), N, dptr(xc), dptr(yc), 2048, float(pupilz_c),
                                                          float(lens.d_sensor), trips_c, flags, dptr(cen), dptr(anyv),
                                                          dptr(mask_c), st)),
            ("trace", lambda: h.sdirt_trace(handle, 0, K, 0, trips_t, flags, ray.c_rays(), M, dptr(mask_t), st)),
            ("propagate_to", lambda: h.sdirt_propagate_to(float(lens.d_sensor), ray.c_rays(), M, st)),
            ("forward_integral", lambda: h.sdirt_forward_integral(ray.c_rays(), S, N, float(lens.pixel_size), ks, dptr(cen),
                                                                  C.byref(dp), flags, dptr(L), dptr(R), st)),
            ("psf_normalize", lambda: h.sdirt_psf_normalize(dptr(LR), 2 * N, ks, st)),
        ]

        # the same work with the two fused entries: trace2sensor in one pass, grids normalised out of the LDS tiles
        # (everything allocated before anything is timed: an allocation idles the GPU long enough to cost it its clock)
        ray2 = Ray.empty((S, N), 0.589, dev)
        L_two_step = torch.empty_like(L)
        calls_fused = [
            calls[0], calls[1],
            ("trace2sensor", lambda: h.sdirt_trace2sensor(handle, trips_t, flags, float(lens.d_sensor), ray.c_rays(), ray2.c_rays(),
                                                          M, dptr(mask_t), st)),
            ("forward_integral_normalized", lambda: h.sdirt_forward_integral(ray2.c_rays(), S, N, float(lens.pixel_size), ks,
                                                                             dptr(cen), C.byref(dp), flags | _lib.PSF_NORMALIZE,
                                                                             dptr(L), dptr(R), st)),
        ]

        def chain(ev=None, which=None):
            for name, fn in (which or calls):
                if ev is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                _lib.check(fn())
                if ev is not None:
                    e1.record(stream)
                    ev.setdefault(name, []).append((e0, e1))
        for _ in range(max(args.warmup, 2)):
            mask_t.zero_(); mask_c.zero_()
            chain()
        torch.cuda.synchronize(dev)
        # the tables the timed chain runs are the reference's for this batch
        from sdirt_amd import newton
        for tab, m in ((t_trace, mask_t), (t_cen, mask_c)):
            ok, _ = newton.verify(tab, lens._read_masks(m), list(range(K)), lens._curved())
            assert ok, "staged chain: speculated Newton trip table is not the reference's"
        steps = max(args.steps, 3)
        ev = {}
        t0 = time.perf_counter()
        for _ in range(steps):
            chain(ev)
        torch.cuda.synchronize(dev)
        wall = (time.perf_counter() - t0) / steps * 1e3
        ms = {k: float(np.mean([a.elapsed_time(b) for a, b in v])) for k, v in ev.items()}
        fused = None
        if getattr(args, "staged_chain", "both") == "both":
            L_two_step.copy_(L)
            for _ in range(max(args.warmup, 2) + 1):         # first touch of the second bundle, first launches of two kernels
                chain(which=calls_fused)
            torch.cuda.synchronize(dev)
            # (one ulp at most: the float64 sums of a tile meet in arrival order before they are rounded to fp32)
            assert float((L - L_two_step).abs().max()) <= 1.2e-7, "fused calls: PSFs differ from the call-by-call chain"
            ev2 = {}
            t0 = time.perf_counter()
            for _ in range(steps):
                chain(ev2, calls_fused)
            torch.cuda.synchronize(dev)
            wall2 = (time.perf_counter() - t0) / steps * 1e3
            ms2 = {k: float(np.mean([a.elapsed_time(b) for a, b in v])) for k, v in ev2.items()}
            # sample -> chief centre -> sdirt_trace2sensor -> sdirt_forward_integral(SDIRT_PSF_NORMALIZE): the same PSFs
            # (checked above) with two passes over memory less
            fused = {"ms_per_step": wall2, "rays_per_s": M / (wall2 * 1e-3), "kernels_ms": ms2}
        grids = 2 * N * ks * ks * 4
        alg = {"sample_rays": 28 * M + 12 * N + 8 * S,
               "trace": 56 * M,
               "propagate_to": 36 * M,
               "forward_integral": 20 * M + 8 * N + grids,
               "psf_normalize": 2 * grids}
        kern = {}
        carried = pmc_counters(f"staged_ks{ks}")
        for name, t in ms.items():
            k = {"ms": t}
            if name in alg:
                gbs = alg[name] / (t * 1e-3) / 1e9
                k.update({"algorithmic_bytes": alg[name], "achieved_GBps": gbs, "frac_of_hbm_peak": gbs / HBM_PEAK_GBS})
                c = ((carried or {}).get("kernels") or {}).get(KERNEL_OF[name])
                nl = 1                                            # (L and R are normalised by one launch over 2 N tiles)
                if nl > 1:
                    k["launches"] = nl
                if c and c.get("hbm_bytes_per_dispatch_mean_last3") and (carried.get("n_points"), carried.get("spp")) == (N, S):
                    # HBM bytes of this kernel from the committed rocprofv3 --pmc passes of this command
                    # (2 x FETCH_SIZE + WRITE_SIZE, MI355X_MICROARCH.md), and its rocprofv3 average duration
                    k.update({"traffic": nl * c["hbm_bytes_per_dispatch_mean_last3"],
                              "traffic_over_algorithmic": nl * c["hbm_bytes_per_dispatch_mean_last3"] / alg[name],
                              # median: the trace's table-discovery launch (10 trips on every surface) is in the average
                              "rocprof_median_ms": nl * c["median_us"] / 1e3, "rocprof_avg_ms": nl * c["avg_us"] / 1e3,
                              "traffic_stale": bool(carried["stale"]),
                              "traffic_file": carried["file"]})
            kern[name] = k
        kern["trace"]["bound"] = kern["chief_center"]["bound"] = "valu"
        out[f"ks{ks}"] = {"ms_per_step": wall, "rays_per_s": M / (wall * 1e-3), "psfs_per_s": N / (wall * 1e-3),
                          "kernels": kern, "sum_of_kernels_ms": float(sum(ms.values())), "fused_calls": fused}
    first = out[f"ks{ks_list[0]}"]
    dom = max(("sample_rays", "propagate_to", "forward_integral"), key=lambda k: first["kernels"][k]["ms"])
    kd = first["kernels"][dom]
    res = {"metric": f"rays/sec rf50mm staged SoA pipeline {ks_list[0]}x{ks_list[0]} DP-PSF @{S}spp",
           "value": first["rays_per_s"], "unit": "rays/s", "n_gpus": 1, "steps": max(args.steps, 3), "warmup": max(args.warmup, 2),
           "ms_per_step": first["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"rf50mm, {N} points of the config-2 volume (every {16384 // N}th) x {S} spp (+2048 chief-ray rays/point), "
                                  "rays staged in HBM as SoA [spp, N]: sample -> chief centre -> trace -> propagate -> "
                                  "forward_integral -> normalise, L+R grids", "name": "staged", "points_per_gpu": N, "spp": S,
                      "ks": list(ks_list), "newton_trip_policy": "reference (tables verified before the timed steps)"},
           "staged": out,
           "roofline": {"bound": "hbm", "achieved": kd["achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": kd["frac_of_hbm_peak"], "traffic": kd.get("traffic"), "kernel": dom,
                        "algorithmic_bytes_per_launch": kd["algorithmic_bytes"],
                        "note": "the slowest of the chain's HBM-bound kernels; k_trace and k_chief_center are bound by vector "
                                "ALU time like the fused kernel (per-kernel figures under `staged`)"}}
    if emit:
        emit_line(res)
    return res


def bench_sweep(args, emit=True, lens=None):
    """--workload sweep: the COMPUTE side of strong scaling, on the one GPU a box of this pool has.  Config 2 cut to
    16384 / 8192 / 4096 / 2048 points per step (what a rank renders at world 1 / 2 / 4 / 8 under `--scaling strong`),
    each through the loop `--gpus N` runs (VolumeLoop with force=True): pupil broadcast, deferred trip check behind a
    mask all-reduce, the shard rendered into its [n, 2, ks, ks] block and that block all-gathered on the comm stream --
    every collective really issued, on a world-1 RCCL process group (sdirt_amd.dist.FORCE_COLLECTIVES).  What a 1-GPU box
    cannot show is the xGMI side: a rank of an 8-GPU run RECEIVES 7 blocks where this stand-in copies its own.
    Every k-th point of the volume (all depth planes, all field positions), so that the batch-global Newton trip
    tables are the whole volume's.  compute_efficiency = (ms of the plain 16384-point single-GPU step x n / 16384) /
    ms_per_step: 1.0 = a rank of an N-GPU run would take exactly 1/N of the single-GPU step (marginal cost of a step,
    from two region lengths; `fences_ms` is what the barrier + synchronisation pair around a timed region adds)."""
    import socket
    import torch.distributed as dist
    from sdirt_amd import dist as sd
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback exists)"
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    own_group = not dist.is_initialized()
    if own_group:
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    sd.FORCE_COLLECTIVES = True
    try:
        if lens is None:
            lens = build_lens(dev)
        pts_all = volume_points(1, "c2").to(dev)
        n_full = pts_all.shape[0]
        steps = max(60, args.steps)
        ks, spp = WORKLOADS["c2"]["ks"], WORKLOADS["c2"]["spp"]
        rows = {}

        def run(points, force, label, streams=1):
            loop = VolumeLoop(lens, points, points.shape[0], 1, dev, ks, spp, gather=force, force=force, streams=streams)
            try:
                loop.step()
                loop.settle()                                    # trip-table discovery for this batch
                for _ in range(10):
                    loop.step()
                loop.fence()
                lens.kernel_events = {}
                del loop.gather_events[:]
                r0 = lens.trips.relaunches
                # two region lengths: the slope is the cost of a step, the intercept what the two fences of a timed
                # region cost (pipeline drain, the last steps' verification and gathers, barrier, device synchronisation)
                dt_short = loop.timed(steps // 4)
                loop.t_step = loop.t_wait = 0.0
                dt = loop.timed(steps)
                marginal = (dt - dt_short) / (steps - steps // 4) * 1e3
                ev, lens.kernel_events = lens.kernel_events, None
                k_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in ev["psf_lr_centered"]]))
                row = {"points_per_step": points.shape[0], "steps": steps, "ms_per_step": dt / steps * 1e3,
                       "ms_per_step_marginal": marginal, "fences_ms": dt * 1e3 - marginal * steps, "kernel_ms": k_ms,
                       "gpu_idle_us_per_step": (marginal - k_ms) * 1e3,
                       # wall time the host spends enqueueing a step (draw, upload, launches, collectives' host side,
                       # verification arithmetic), without the time it sits blocked waiting for a result
                       "host_us_per_step": (loop.t_step - loop.t_wait) / steps * 1e6,
                       "rays_per_s": points.shape[0] * spp * steps / dt,
                       "relaunches_in_timed_region": lens.trips.relaunches - r0,
                       "trip_tables": {"psf": [int(v) for v in lens.trips.cache[("psf", 0.589, lens.precision)]],
                                       "center": [int(v) for v in lens.trips.cache[("center", lens.precision)]]}}
                if loop.gather_events:
                    row["gather_ms"] = float(np.mean([a_.elapsed_time(b_) for a_, b_ in loop.gather_events]))
                    row["gather_block_mb"] = points.shape[0] * 2 * ks * ks * 4 / 1e6
                rows[label] = row
                return row
            finally:
                loop.close()
                del loop
                torch.cuda.empty_cache()
        import gc
        base = run(pts_all, False, "single_gpu_loop_16384")       # the headline's own loop: no process group in the way
        gc.collect()
        gc.freeze()
        def rate(row):
            # marginal step against marginal step: what a long run converges to; `..._incl_fences`: this 100-step region
            row["compute_efficiency"] = base["ms_per_step_marginal"] * (row["points_per_step"] / n_full) / row["ms_per_step_marginal"]
            row["compute_efficiency_incl_fences"] = base["ms_per_step"] * (row["points_per_step"] / n_full) / row["ms_per_step"]
            row["kernel_efficiency"] = base["kernel_ms"] * (row["points_per_step"] / n_full) / row["kernel_ms"]
            row["trip_tables_equal_full_batch"] = row.pop("trip_tables") == base["trip_tables"]
        for k in (1, 2, 4, 8):
            for streams in (1, 2):
                # streams = 2: consecutive steps on alternating render streams -- the next step's workgroups fill the
                # low-occupancy end of the previous launch (`kernel_ms` then spans two overlapping launches)
                row = run(pts_all[::k].contiguous(), True, f"world{k}_shard_{n_full // k}" + ("" if streams == 1 else "_two_streams"), streams)
                row["as_rank_of_world"], row["render_streams"] = k, streams
                rate(row)
        rate(run(pts_all, False, "single_gpu_loop_16384_two_streams", 2))
    finally:
        sd.FORCE_COLLECTIVES = False
        if own_group:
            dist.destroy_process_group()
    last = rows[f"world8_shard_{n_full // 8}"]
    res = {"metric": "rays/sec rf50mm 65x65 DP-PSF @4096spp, strong-scaling compute side on one GPU (2048-point step of a rank of 8)",
           "value": last["rays_per_s"], "unit": "rays/s", "n_gpus": 1, "steps": steps, "warmup": 10,
           "ms_per_step": last["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic",
           "config": {"workload": "config 2 (rf50mm 32x32x16, 4096 spp, 65x65 L+R) cut to every k-th point, k = 1, 2, 4, 8: the step of "
                                  "one rank of a k-GPU strong-scaling run, all collectives issued on a world-1 RCCL group",
                      "name": "sweep", "backend": "nccl (RCCL), world 1",
                      "not_measured": "xGMI: a rank of world k receives k - 1 blocks; here the gather copies the rank's own"},
           "shard_sweep": rows}
    if emit:
        emit_line(res)
    return res


def bench_c5(args, emit=True):
    """BASELINE config 5 on one GPU, end to end (2_dfdp_net.py's image simulation, then its depth network): a synthetic
    512 x 768 RGB-D frame (NYUv2 is not in the reference's repository) -> PSFNet.render (psfnet.py:645-714: per-pixel
    L/R kernels from the PSF network, per-pixel convolution; ks 21, full-size MLP with seeded weights -- the
    reference's checkpoints are missing) -> DfDPNet forward under fp16 autocast (dfdp/dddnet/dddnet.py:122-152).
    MIOpen's find mode (torch.backends.cudnn.benchmark) is OFF: the convolutions run with MIOpen's immediate-mode picks."""
    from sdirt_amd.dfdp import DfDPNet
    from sdirt_amd.psfnet import PSFNet
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback exists)"
    dev = torch.device("cuda", 0)
    find_mode = bool(torch.backends.cudnn.benchmark)
    H, W, ks = 512, 768, 21
    torch.manual_seed(0)
    m = PSFNet(os.path.join(ROOT, "sdirt_amd", "data", "rf50mm.json"), sensor_res=(H, W), kernel_size=ks, device=dev)
    m.refocus(-1000 + m.d_sensor)
    with torch.no_grad():
        m.psfnet.net[-2].bias.add_(0.02)        # seeded weights: keep the raw kernels away from an all-zero sum
    g = torch.Generator(device=dev).manual_seed(0)
    img = torch.rand(1, 3, H, W, device=dev, generator=g)
    depth = -(500 + 4500 * torch.rand(1, 1, H, W, device=dev, generator=g))       # 0.5 ... 5 m
    foc = torch.tensor([-1000.0], device=dev)
    torch.manual_seed(1)
    net = DfDPNet().to(dev).eval()
    stream = torch.cuda.current_stream(dev)
    ev = {"render": [], "dfdp_forward": []}

    def chain(record=False):
        with torch.no_grad():
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if record else None
            if record:
                e[0].record(stream)
            pair = m.render(img, depth, foc)
            if record:
                e[1].record(stream)
            left, right = pair[:, :3].contiguous(), pair[:, 3:].contiguous()
            with torch.autocast("cuda", dtype=torch.float16):
                disp = net(left, right)
            if record:
                e[2].record(stream)
                ev["render"].append((e[0], e[1]))
                ev["dfdp_forward"].append((e[1], e[2]))
        return disp
    t0 = time.perf_counter()
    for _ in range(max(args.warmup, 3)):
        disp = chain()
    torch.cuda.synchronize(dev)
    warm_s = time.perf_counter() - t0
    steps = max(args.steps, 5)
    t0 = time.perf_counter()
    for _ in range(steps):
        disp = chain(True)
    torch.cuda.synchronize(dev)
    wall = (time.perf_counter() - t0) / steps * 1e3
    assert bool(torch.isfinite(disp).all())
    ms = {k: float(np.mean([a.elapsed_time(b) for a, b in v])) for k, v in ev.items()}
    res = {"metric": "frames/sec config 5: RGB-D 512x768 -> PSFNet.render (DP pair) -> DfDP net forward, 1 GPU",
           "value": 1e3 / wall, "unit": "frames/s", "n_gpus": 1, "steps": steps, "warmup": max(args.warmup, 3),
           "ms_per_step": wall, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16",
           "data": "synthetic",
           "config": {"workload": "synthetic RGB-D frame 1x3x512x768 (depth U[0.5, 5] m, focus 1 m) -> PSFNet.render, ks 21, full-size PSF "
                                  "network (seeded weights) -> DfDPNet forward (fp16 autocast), rf50mm", "name": "c5",
                      "miopen_find_mode": find_mode, "first_calls_s": warm_s},
           "kernels_ms": ms}
    if emit:
        emit_line(res)
    return res


def quick_volume(workload, steps, device):
    """K steps of another PSF-volume workload (WORKLOADS) with the stepping of the headline loop -- calls kept in
    flight, Newton trip check of step i under step i + 1 -- for the `also` block of the default line."""
    wl = WORKLOADS[workload]
    lens = build_lens(device, wl["lens"], wl["sensor_z"])
    gz = 8 if workload.startswith("c3") else wl["grid_z"]
    pts = volume_points(1, workload).to(device)
    n, ks, spp = pts.shape[0], wl["ks"], wl["spp"]
    bufs = [tuple(torch.empty((n, ks, ks), dtype=torch.float32, device=device) for _ in range(2)) for _ in range(3)]
    lens.psf_lr(pts, ks=ks, spp=spp, dp=DP, out=bufs[0])             # trip-table discovery
    lens.psf_lr(pts, ks=ks, spp=spp, dp=DP, out=bufs[0])
    torch.cuda.synchronize(device)
    lens.kernel_events = {}
    r0 = lens.trips.relaunches
    pend = []
    t0 = time.perf_counter()
    for i in range(steps):
        pend.append(lens.psf_lr(pts, ks=ks, spp=spp, dp=DP, out=bufs[i % 3], defer=True))
        if len(pend) > 2:
            pend.pop(0).wait()
    for p_ in pend:
        p_.wait()
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    ev, lens.kernel_events = lens.kernel_events, None
    k_ms = {k: float(np.mean([a.elapsed_time(b) for a, b in v])) for k, v in ev.items()}
    return {"metric": f"rays/sec {wl['lens']} {ks}x{ks} DP-PSF @{spp}spp", "value": n * spp * steps / dt, "unit": "rays/s",
            "steps": steps, "ms_per_step": dt / steps * 1e3, "kernels_ms": k_ms,
            "config": {"workload": wl["desc"].format(gz=gz) + f", {n} points, {spp} spp, {ks}x{ks} L+R", "name": workload,
                       "relaunches_in_timed_region": lens.trips.relaunches - r0}}


def also_block(args, lens, device):
    """The other driver-timed lines of the default run: the staged SoA chain (HBM-bound kernels; call by call and through
    the fused entries), the per-pixel PSF convolution f1 (HBM-bound), config 4 (rf35mm, the second prescription), one
    GPU's share of config 3 (8192 points x 8192 spp, 21 x 21), the reference's own timing harness (tcp), config 5 end to
    end (c5) -- five steps each -- and the strong-scaling compute side (shard_sweep: 50 steps per shard size)."""
    import copy
    q = copy.copy(args)
    q.steps, q.warmup, q.sustain_seconds = 20, 2, 0.0
    out = {}
    # the staged chain's steps are 3 ms: warmed for ~0.1 s first -- the clock of a chip that has just been idle (set-up,
    # allocations) dips for some tens of ms shortly after work resumes (tools/clock_ramp.py), longer than five such steps
    qs = copy.copy(q)
    qs.steps, qs.warmup = 10, 40
    qs.staged_chain = "both"           # the call-by-call chain and the one through the fused entries (`fused_calls`)
    for name, fn in (("staged", lambda: bench_staged(qs, emit=False, lens=lens)),
                     ("f1", lambda: bench_f1(q, emit=False)),
                     ("c4", lambda: quick_volume("c4", 20, device)),
                     ("c3", lambda: quick_volume("c3", 20, device)),
                     ("tcp", lambda: bench_tcp(q, emit=False)),
                     ("c5", lambda: bench_c5(q, emit=False)),
                     # last: it opens (and closes) a world-1 RCCL process group in this process
                     ("shard_sweep", lambda: bench_sweep(q, emit=False, lens=lens))):
        t0 = time.perf_counter()
        try:
            r = fn()
            keep = ("metric", "value", "unit", "steps", "ms_per_step", "config", "roofline", "kernels_ms", "staged",
                    "value_pcie_inclusive", "reference_harness", "shard_sweep")
            out[name] = {k: r[k] for k in keep if k in r}
        except Exception as e:       # a broken side line must not cost the headline
            out[name] = {"error": f"{type(e).__name__}: {e}"}
        out[name]["wall_s"] = time.perf_counter() - t0
        torch.cuda.synchronize(device)
    return out


def bench_tcp(args, emit=True):
    """The one workload the reference itself times: PSFNet.time_compare_psf (psfnet.py:570-586) -- 24576
    random points, 4096 spp, ks 21, the PSFs copied to the host inside the timed span.  `value` is the
    device-resident rate (inputs and outputs in HBM, as every other line of this file);
    `value_pcie_inclusive` is the reference's own span, host copy included."""
    from sdirt_amd.psfnet import PSFNet
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback exists)"
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    np.random.seed(0)
    m = PSFNet(os.path.join(ROOT, "sdirt_amd", "data", "rf50mm.json"), sensor_res=(512, 768), kernel_size=21,
               device=dev)
    m.refocus(-1000 + m.d_sensor)
    n, spp, ks = 512 * 768 // 16, 4096, 21
    for _ in range(max(args.warmup, 1)):
        m.time_compare_psf(verbose=False)
    t_trace, t_net = zip(*[m.time_compare_psf(verbose=False) for _ in range(args.steps)])
    inp = torch.rand(n, 3)
    inp[:, 2] = m.z2depth(inp[:, 2])
    ind = inp.to(dev)
    out = tuple(torch.empty((n, ks, ks), device=dev) for _ in range(2))
    fn = lambda: m.psf_lr(ind, ks=ks, spp=spp, out=out, want_r=False, _default_r_zero=True)
    wall, kern = _hip_ms(fn, args.steps, 2, dev)
    res = {"metric": "rays/sec PSFNet.time_compare_psf: 24576 random points, 4096 spp, ks 21 (rf50mm)",
           "value": n * spp / (wall * 1e-3), "unit": "rays/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": wall, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
           "data": "synthetic",
           "value_pcie_inclusive": n * spp / float(np.mean(t_trace)),
           "reference_harness": {"ray_tracing_seconds": float(np.mean(t_trace)), "network_seconds": float(np.mean(t_net)),
                                 "what": "wall clock of the two spans of psfnet.py:570-586 as the reference prints them: "
                                         "psf(...).to('cpu') (42.5 MB to the host) and pred(...).to('cpu') for a 128x192 field"},
           "config": {"workload": "PSFNet.time_compare_psf: 24576 random points x 4096 spp (+2048 chief-ray rays/point), "
                                  "21x21 L PSFs (param_list=None), lambda 0.589um, focus 1 m", "name": "tcp",
                      "points_per_gpu": n, "spp": spp, "ks": ks},
           "kernels_ms": {"psf_lr synchronous call (events)": kern}}
    if emit:
        emit_line(res)
    return res


class VolumeLoop:
    """The stepping of one rank of a PSF-volume render: `step()` enqueues one psf_lr call over the rank's points with
    the speculated Newton trip tables (defer=True) and keeps DEPTH calls in flight; `settle()` verifies them in order
    and -- when the step asked for it -- sends the now final shard through the all-gather on the comm stream;
    `timed(k)` = k steps between two fences (barrier + device synchronisation), MAX over ranks.

    world > 1 (or force=True: the single-rank stand-in of `--workload sweep`, every collective issued on a world-1
    process group): the ranks share the pupil sample set (48 KB broadcast per step) and the batch-global trip check
    (mask all-reduce, sdirt_amd/dist.py) and the shard is gathered to every rank.

    A rank renders into ONE [width, 2, ks, ks] block per step -- point n's left grid at [n, 0], its right grid at
    [n, 1] (SDIRT_PSF_INTERLEAVED) -- which is also what the gather sends: one collective per step, no staging copy
    (ShardedPSF.shard_buffer / gather: the library path and the benchmarked path are the same)."""
    DEPTH = 8        # calls kept in flight (kernel enqueued, Newton trip check pending): ~80 ms of queued work

    def __init__(self, lens, points_local, n_total, world, device, ks, spp, gather, force=False, streams=1):
        import torch.distributed as dist
        from sdirt_amd import dist as sd
        self.lens, self.points, self.n_total, self.world, self.device = lens, points_local, n_total, world, device
        self.ks, self.spp, self.gather, self.dist, self.sd = ks, spp, gather, dist, sd
        self.n_local = points_local.shape[0]
        self.multi = world > 1 or force
        self.gather_group = self.comm_stream = self.sharded = None
        if self.multi:
            self.sharded = sd.ShardedPSF.from_lens(lens, ks, dp=DP)
            # the all-gather of step i runs on its own stream underneath the kernels of step i+1
            # (double-buffered: xGMI copy engines / RCCL channels vs. VALU-bound compute)
            self.comm_stream = torch.cuda.Stream(device)
            # its own communicator: PyTorch runs all collectives of one process group on one internal
            # stream, so a gather issued on the default group would hold back the next step's small
            # pupil broadcast (and with it the next kernel) until 4 GB have moved
            self.gather_group = dist.new_group() if gather else None
            # ... and the pupil broadcast of step i+1 on a third one, on a side stream: it depends on nothing that is
            # queued, so it runs beside step i's kernel (on the default group it would queue behind step i's mask
            # all-reduce, i.e. behind step i's kernel: 0.16-0.35 ms of idle GPU per step, `--workload sweep`)
            self.pupil_group = dist.new_group()
            self.pupil_stream = torch.cuda.Stream(device)
        self.width = max(b_ - a_ for a_, b_ in sd.shard_bounds(n_total, world))
        self.gather_buf = [torch.empty((world * self.width, 2, ks, ks), dtype=torch.float32, device=device)
                           for _ in range(2)] if (gather and self.multi) else None
        self.step_no = 0
        # output blocks are owned by the caller and re-used (the previous steps' PSFs may still be
        # feeding the all-gather / the consumer while the next step renders)
        self.out_bufs = [torch.zeros((self.width, 2, ks, ks), dtype=torch.float32, device=device)
                         for _ in range(self.DEPTH + 1)]
        self.gather_done = [None] * (self.DEPTH + 1)   # per output block: event of the last gather reading it
        self.gathers = 0
        self.gather_events = []                        # (start, end) on the comm stream, one pair per step's all-gather
        self.in_flight = []                            # (PendingPSF, out, slot, ready event | None)
        self.t_step = self.t_wait = 0.0                # wall seconds inside step() / of them blocked in PendingPSF.wait()
        self.in_step = False
        # streams = 2: consecutive steps alternate between two render streams.  Steps are independent (their own
        # output block, pupil set and control block), and on one in-order stream step i+1 cannot start before the LAST
        # workgroup of step i has finished -- the end of a launch runs at falling occupancy (0.13 ms of a
        # 16384-point step, 10 % of a 2048-point one); on two streams the next step's workgroups fill it.
        self.render_streams = [torch.cuda.Stream(device) for _ in range(streams)] if streams > 1 else None
        torch.cuda.synchronize(device)                 # buffers above are ready whatever stream uses them first

    def use_streams(self, n):
        """Switch between one render stream (the caller's) and n alternating ones; between two fences only."""
        self.fence()
        self.render_streams = [torch.cuda.Stream(self.device) for _ in range(n)] if n > 1 else None

    def close(self):
        """Hand the lens back to single-rank use."""
        if self.multi:
            self.lens.mask_reduce = None

    def settle(self, keep=0):
        """Newton trip checks (lens.psf_lr(defer=True)) of all but the `keep` newest steps, each
        followed -- when the step asked for it -- by the all-gather of its now final shard on
        the comm stream.  The host stays `keep` kernels ahead of the GPU: the GPU renders step
        i+1 while the host verifies step i and RCCL moves step i's PSFs, and a descheduled host
        thread (the boxes are shared) does not leave the GPU idle."""
        lens, device = self.lens, self.device
        while len(self.in_flight) > keep:
            pend, out, slot, ready = self.in_flight.pop(0)
            r0 = lens.trips.relaunches
            t_w = time.perf_counter()
            pend.wait()
            if self.in_step:
                self.t_wait += time.perf_counter() - t_w
            if ready is None:
                continue
            if lens.trips.relaunches != r0:          # re-rendered: the shard is ready later
                ready = torch.cuda.Event()
                ready.record(torch.cuda.current_stream(device))
            buf = self.gather_buf[self.gathers % 2]
            self.gathers += 1
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ready)
                g0 = torch.cuda.Event(enable_timing=True)
                g0.record(self.comm_stream)
                self.sharded.gather(out, self.n_total, out=buf, group=self.gather_group)   # ONE collective
                done = torch.cuda.Event(enable_timing=True)
                done.record(self.comm_stream)
                self.gather_events.append((g0, done))
            self.gather_done[slot] = done

    def step(self, gather=None):
        t0 = time.perf_counter()
        self.in_step = True
        try:
            return self._step(gather)
        finally:
            self.in_step = False
            self.t_step += time.perf_counter() - t0

    def _step(self, gather=None):
        if self.render_streams is None:
            return self._step_on_current_stream(gather)
        with torch.cuda.stream(self.render_streams[self.step_no % len(self.render_streams)]):
            return self._step_on_current_stream(gather)

    def _step_on_current_stream(self, gather=None):
        gather = self.gather if gather is None else gather
        lens, device, n_local = self.lens, self.device, self.n_local
        slot = self.step_no % (self.DEPTH + 1)
        out = self.out_bufs[slot]
        self.step_no += 1
        if not self.multi:
            self.in_flight.append((lens.psf_lr(self.points, ks=self.ks, spp=self.spp, dp=DP, out=out[:n_local], defer=True),
                                   out, slot, None))
            self.settle(keep=self.DEPTH)
            return out
        if self.gather_done[slot] is not None:
            # an earlier gather may still read this block on the comm stream
            torch.cuda.current_stream(device).wait_event(self.gather_done[slot])
            self.gather_done[slot] = None
        pupil = self.sd.broadcast_pupil_points(lens, self.spp, group=self.pupil_group, stream=self.pupil_stream)
        pend = self.sharded.render(self.points, pupil, out[:n_local], defer=True)
        ready = None
        if gather:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(device))
        self.in_flight.append((pend, out, slot, ready))
        self.settle(keep=3 if gather else self.DEPTH)   # gathers trail by three steps: a host hiccup on one rank stalls nobody
        return out

    def fence(self):
        self.settle()
        torch.cuda.synchronize(self.device)          # all streams of the device, the gather one too
        if self.multi:
            self.dist.barrier()
            torch.cuda.synchronize(self.device)

    def timed(self, k, gather=None):
        """k steps between two fences; wall time = MAX over ranks."""
        self.fence()
        t0 = time.perf_counter()
        for _ in range(k):
            self.step(gather)
        self.fence()
        dt = time.perf_counter() - t0
        if self.multi:
            tmax = torch.tensor([dt], dtype=torch.float64, device=self.device)
            self.dist.all_reduce(tmax, op=self.dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-gather", action="store_true",
                    help="N > 1: skip the RCCL all-gather of the PSF shards (by default every rank "
                         "ends each step with the whole PSF volume; `value_no_gather` is reported "
                         "beside `value` either way)")
    ap.add_argument("--gather", action="store_true", help="(default; kept for older scripts)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="N > 1: weak (default) = every rank renders its own 16384-point slab of a volume N times as "
                         "deep; strong = the ONE volume of the workload (c2 / c4: 16384 points, c3: 65536) cut into N "
                         "contiguous shards")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true",
                    help="default c2 run at N = 1: skip the `also` block (staged SoA chain, f1, c4: a few steps each)")
    ap.add_argument("--sustain-seconds", type=float, default=20.0,
                    help="after the K timed steps, keep stepping for this long and report "
                         "ms_per_step_sustained (0 = skip); long enough for a 5-second GPU-activity "
                         "sampler to see the GPU phase of the run")
    ap.add_argument("--workload", choices=sorted(WORKLOADS) + list(EXTRA_WORKLOADS), default="c2",
                    help="c2 (default, the headline) / c3 / c3k65 / c4: PSF-volume renders; f1: per-pixel DP-PSF "
                         "convolution of a 512x768 frame (render_psf.py:120-155); tcp: the reference's own "
                         "timing harness PSFNet.time_compare_psf (psfnet.py:570-586); staged: the reference's own "
                         "call sequence sample -> trace -> propagate -> forward_integral on SoA rays in HBM; sweep: config 2 "
                         "cut to the step of one rank of a 1 / 2 / 4 / 8-GPU strong-scaling run, collectives on a world-1 RCCL "
                         "group; c5: config 5 end to end (RGB-D frame -> PSFNet.render -> DfDP net forward)")
    ap.add_argument("--staged-ks", default="65,21", help="--workload staged: the grid sizes to run the chain for")
    ap.add_argument("--staged-chain", choices=("both", "calls"), default="both",
                    help="--workload staged: `calls` times the call-by-call chain only (the profiling recipe: one kernel "
                         "per name), `both` also the chain through sdirt_trace2sensor / SDIRT_PSF_NORMALIZE")
    args = ap.parse_args()
    if not (args.gpus > 1 and "WORLD_SIZE" not in os.environ):      # (the self-launching parent relays its children's output)
        claim_stdout()
    if args.workload in EXTRA_WORKLOADS:
        assert args.gpus == 1, f"--workload {args.workload} is a single-GPU measurement"
        return {"f1": bench_f1, "tcp": bench_tcp, "staged": bench_staged, "sweep": bench_sweep, "c5": bench_c5}[args.workload](args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    global KS, SPP, GRID_Z
    wl = WORKLOADS[args.workload]
    KS, SPP = wl["ks"], wl["spp"]
    # c3 is a 65536-point grid meant for 8 GPUs: per-GPU slab = 8 depth planes
    GRID_Z = wl["grid_z"] if not args.workload.startswith("c3") else 8

    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback exists)"
    # SDIRT_BENCH_BACKEND=gloo is a DRY-RUN aid for boxes with fewer GPUs than ranks (all ranks
    # share the visible devices round-robin, collectives go through gloo): it exercises the
    # multi-rank control flow, its numbers mean nothing.  The driver's runs use nccl (= RCCL).
    backend = os.environ.get("SDIRT_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from sdirt_amd import dist as sd
    lens = build_lens(device, wl["lens"], wl["sensor_z"])
    strong = args.scaling == "strong"
    # weak: the volume grows with the ranks (16 -- c3: 8 -- depth planes per GPU); strong: the workload's own volume
    # (c3 is stated for a node of 8: 64 planes) whatever the number of ranks
    points_all = volume_points((8 if args.workload.startswith("c3") else 1) if strong else world, args.workload)
    n_total = points_all.shape[0]
    a, b = sd.shard_bounds(n_total, world)[rank]
    points_local = points_all[a:b].to(device)
    n_local = b - a
    gather_default = world > 1 and not args.no_gather
    # N > 1: consecutive steps alternate between two render streams (the next step's workgroups fill the low-occupancy end
    # of the previous launch: 3.5 % of a 16384-point step, 7 % of a 2048-point one with the collectives in the loop,
    # `--workload sweep`).  N = 1 keeps ONE stream: the headline's kernel time is then the time of a launch that has
    # the chip to itself, which is what `roofline` and the committed rocprofv3 summaries are about.
    n_streams = 2 if world > 1 else 1
    loop = VolumeLoop(lens, points_local, n_total, world, device, KS, SPP, gather_default, streams=n_streams)
    step, settle, timed = loop.step, loop.settle, loop.timed
    gather_events, gather_group, width, out_bufs = loop.gather_events, loop.gather_group, loop.width, loop.out_bufs

    # one-off initialisation, like loading the library: the first call on a lens discovers the
    # Newton trip tables (a launch with 10 trips everywhere, then the verified table); they are
    # lens state from then on.  Done before the W warm-up steps so that W = 0 still times K
    # steady-state steps.
    step(gather_default)
    settle()
    # Everything the run needs is built: park the interpreter's heap in the permanent generation.  A full collection of
    # torch's heap takes 40-55 ms (measured: it strikes ~150 ms after the first launch, i.e. inside the timed window of
    # a --warmup 5 --steps 20 run, idles the GPU and costs it its clock: tools/clock_ramp.py); after the freeze the
    # collector only ever walks what the steps themselves allocate.
    import gc
    gc.collect()
    gc.freeze()
    for _ in range(args.warmup):
        step(gather_default)
    lens.kernel_events = {}
    relaunch0 = lens.trips.relaunches
    del gather_events[:]
    dt = timed(args.steps, gather_default)
    relaunches = lens.trips.relaunches - relaunch0
    # the all-gathers of the timed steps as the comm stream saw them (start = the shards are final and the previous
    # gather has left the stream, end = every rank's shards have arrived here); MAX over ranks like the wall time
    gather_ms = None
    if gather_default and gather_events:
        gm = torch.tensor([float(np.mean([a_.elapsed_time(b_) for a_, b_ in gather_events]))], dtype=torch.float64, device=device)
        dist.all_reduce(gm, op=dist.ReduceOp.MAX)
        gather_ms = float(gm.item())

    ev = lens.kernel_events
    lens.kernel_events = None
    k_ms = {k: float(np.mean([e0.elapsed_time(e1) for e0, e1 in v])) for k, v in ev.items()}
    n_launch = {k: len(v) for k, v in ev.items()}

    # the same K steps without the all-gather (N > 1), and a loop long enough for the clocks to
    # settle (the K-step region lasts a fraction of a second)
    dt_ng = timed(args.steps, False) if gather_default else None
    k_sus = dt_sus = None
    if args.sustain_seconds > 0:
        k_sus = max(args.steps, int(math.ceil(args.sustain_seconds / (dt / args.steps))))
        dt_sus = timed(k_sus, gather_default)

    # N = 1: the same K steps once more with consecutive steps on two alternating streams (what a pipelined consumer of
    # batch after batch gets: the next launch's workgroups fill the low-occupancy end of the previous one)
    dt_two = None
    if world == 1:
        loop.use_streams(2)
        for _ in range(max(args.warmup, 2)):
            step(False)
        dt_two = timed(args.steps, False)
        loop.use_streams(1)

    if rank == 0:
        rays = n_total * SPP * args.steps
        # algorithmic HBM bytes of ONE k_psf_lr launch: read the points,
        # centres and pupil samples once, write the L and R tiles once.
        alg_bytes = n_local * (12 + 8) + (SPP + 2048) * 8 + 2 * n_local * KS * KS * 4
        dom = "psf_lr_centered" if "psf_lr_centered" in k_ms else "psf_lr"
        k_events = k_ms[dom]
        if n_streams > 1:
            # two launches overlap on the chip: an event pair spans both.  What a launch costs the step is the step itself
            # (without the gather): that is the duration the roofline figures and `gather_bound` are computed with.
            k_ms = dict(k_ms)
            k_ms[dom] = min(k_events, (dt_ng if dt_ng is not None else dt) / args.steps * 1e3)
        ach = alg_bytes / (k_ms[dom] * 1e-3) / 1e9
        traffic = valu = None
        counters = pmc_counters(args.workload)
        # algorithmic flops of one launch: ~290 per curved surface (4.5 sag evaluations,
        # normal, refraction, updates) + stop / propagate / splat: 3.4 kflop per primary and 3.3 kflop per
        # chief-ray ray on rf50mm's 11 curved surfaces, the same per-surface figure on rf35mm's 20
        n_curved = sum(1 for c_ in lens._curved() if c_)
        flop_primary, flop_chief = 290 * n_curved + 210, 290 * n_curved + 110
        alg_flops = n_local * (SPP * flop_primary + 2048 * flop_chief)
        ach_tflops = alg_flops / (k_ms[dom] * 1e-3) / 1e12
        valu_flops = {"algorithmic_flops_per_launch": alg_flops, "flop_per_primary_ray": flop_primary,
                      "flop_per_chief_ray": flop_chief, "achieved": ach_tflops, "peak": VALU_PEAK_TFLOPS,
                      "unit": "TFLOP/s", "frac": ach_tflops / VALU_PEAK_TFLOPS,
                      "note": "peak counts an FMA as 2 flops; the parity contract (-ffp-contract=off: the reference's "
                              "unfused torch ops) halves what this instruction stream can reach"}
        if counters:
            traffic = counters.get("k_psf_lr_hbm_bytes_per_launch")
            n_instr = counters.get("k_psf_lr_valu_wave_instructions_per_launch")
            if n_instr:
                # vector-instruction issue: peak = one full-rate wave64 instruction per 2 cycles per SIMD (4 SIMDs per
                # compute unit of THIS device, at the 2.4 GHz peak clock the 157.3 TFLOP/s figure is quoted for);
                # half-rate forms and v_rcp / v_rsq make the reachable figure lower
                n_simd = 4 * int(torch.cuda.get_device_properties(device).multi_processor_count)
                peak = n_simd * 2.4e9 / 2
                n_all = n_instr + (counters.get("salu_wave_instructions_per_launch") or 0) \
                    + (counters.get("smem_instructions_per_launch") or 0)
                clk = counters.get("shader_clock_ghz") or 2.38
                valu = {"wave_instructions_per_launch": n_instr,
                        "all_instructions_per_launch": n_all,
                        "simds": n_simd,
                        "cycles_per_vector_instruction_per_simd": k_ms[dom] * 1e-3 * clk * 1e9 * n_simd / n_instr,
                        "plain_fp32_cycles_per_instruction": 2.25,
                        "achieved_per_s": n_instr / (k_ms[dom] * 1e-3), "peak_per_s": peak,
                        "frac": n_instr / (k_ms[dom] * 1e-3) / peak,
                        # what THIS instruction stream can reach: measured issue costs per SIMD at 8 waves
                        # (profiles/r02/form_bench.txt: v_fma / v_mul / v_add 2.25 cycles, v_rcp / v_rsq / v_sqrt
                        # 8.2) x the counted instructions, over the SIMD cycles the launch had
                        "issue_bound": _issue_bound(n_instr, counters.get("trans_f32_instructions_per_launch"),
                                                    k_ms[dom] * 1e-3 * clk * 1e9 * n_simd),
                        "stale": bool(counters["stale"]),
                        "source": {"kind": "carried: SQ_INSTS_VALU of a separate rocprofv3 --pmc run "
                                           "of this command, divided by THIS run's kernel time",
                                   "file": counters["file"], "collected": counters.get("collected"),
                                   "commit": counters.get("commit"),
                                   "source_hash": counters.get("source_hash"), "source_hash_now": source_hash()}}
        res = {
            "metric": f"rays/sec {wl['lens']} {KS}x{KS} DP-PSF @{SPP}spp", "value": rays / dt,
            "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak",
            "world_size": dist.get_world_size() if world > 1 else 1,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "psfs_per_sec": n_total * args.steps / dt,
            "backend": backend,
            "config": {"workload": wl["desc"].format(gz=(64 if args.workload.startswith("c3") else GRID_Z) if strong else GRID_Z * world)
                                   + (f" cut into {world} shards" if strong and world > 1 else "")
                                   + f", {n_local} points/GPU, {SPP} spp (+2048 chief-ray "
                                   f"rays/point), {KS}x{KS} L+R PSFs, lambda 0.589um, focus 1 m F/4",
                       "name": args.workload,
                       "points_per_gpu": n_local, "spp": SPP, "ks": KS,
                       "parallelism": f"points sharded over {world} GPU(s)"
                                      + ("" if world == 1 else
                                         ", shared pupil samples (48 KB broadcast) and batch-global "
                                         "Newton trip check (mask all-reduce) per step")
                                      + (" + RCCL all-gather of the PSF volume to every rank"
                                         if gather_default else ""),
                       "gather": bool(gather_default),
                       "newton_trip_policy": lens.trip_policy,
                       "relaunches_in_timed_region": relaunches},
            "kernels_ms": k_ms, "kernel_launches": n_launch, "render_streams": n_streams,
            "kernel_event_ms": k_events,
            # what bounds the kernel is its vector ALU time (profiles/r03/k_psf_lr_sites.txt):
            # `valu_flops` is the fraction that says something; achieved / peak / frac / traffic are the
            # mandated HBM figures (0.7 % by construction: 8 algorithmic bytes against 3.4 kflop per ray)
            "roofline": {"bound": "valu", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_stale": bool(counters["stale"]) if counters else None,
                         "valu_flops": valu_flops,
                         "kernel": "k_psf_lr<R,small-r,Lean,CENTER> (chief-ray pass + primary pass "
                                   "of a point in one workgroup)",
                         "algorithmic_bytes_per_launch": alg_bytes, "valu_issue": valu,
                         # the committed rocprofv3 --kernel-trace --stats of this command (carried, like the
                         # counters): its median and its average without the two table-discovery launches a
                         # process starts with are the figures kernels_ms has to agree with
                         "rocprof_kernel_trace": None if not counters else {
                             "file": counters["file"], "stale": bool(counters["stale"]),
                             "median_ms": (counters.get("kernel_trace_median_us") or 0) / 1e3 or None,
                             "avg_steady_ms": (counters.get("kernel_trace_avg_steady_us") or 0) / 1e3 or None,
                             "avg_all_launches_ms": (counters.get("kernel_trace_avg_us") or 0) / 1e3 or None},
                         "note": "scalar-per-ray fp32 math: the kernel is bound by vector-instruction "
                                 "issue by construction (%s VALU instr per traced ray vs %.2f "
                                 "algorithmic bytes), DESIGN.md §3"
                                 % ("%.0f" % (valu["wave_instructions_per_launch"] * 64 / (n_local * (SPP + 2048)))
                                    if valu else "~5 k", alg_bytes / (n_local * (SPP + 2048)))},
        }
        if dt_ng is not None:
            res["value_no_gather"] = rays / dt_ng
            res["ms_per_step_no_gather"] = dt_ng / args.steps * 1e3
            gb = 2 * (n_total - n_local) * KS * KS * 4 / 1e9      # received per rank and step
            compute_ms = k_ms[dom]
            res["gather"] = {"algo": os.environ.get("SDIRT_GATHER_ALGO", "allgather"),
                             "backend": dist.get_backend(gather_group), "world_size": dist.get_world_size(gather_group),
                             "gb_received_per_rank_per_step": gb, "collectives_per_step": 1,
                             "block": f"[{width}, 2, {KS}, {KS}] fp32 per rank, rendered in place (SDIRT_PSF_INTERLEAVED)",
                             "ms": gather_ms, "GBps_received_per_rank": gb / (gather_ms * 1e-3) if gather_ms else None,
                             "compute_ms": compute_ms,
                             # the gather runs on its own stream under the next step's kernel: it costs the step
                             # time only when it takes longer than the kernel it hides under
                             "gather_bound": bool(gather_ms is not None and gather_ms > compute_ms),
                             "what": "ms = mean HIP-event time of one step's all-gather (L and R in one block) on the comm stream in the "
                                     "timed region, MAX over ranks; compute_ms = this rank's kernel time per step"}
        if dt_two is not None:
            res["value_two_streams"] = rays / dt_two
            res["ms_per_step_two_streams"] = dt_two / args.steps * 1e3
        if dt_sus is not None:
            res["ms_per_step_sustained"] = dt_sus / k_sus * 1e3
            res["value_sustained"] = n_total * SPP * k_sus / dt_sus
            res["sustained_steps"] = k_sus
        pupil_last = lens.last_pupil_points
        if world == 1 and args.workload == "c2" and not args.no_also:
            del out_bufs[1:]                    # 4.4 GB of PSF buffers the side lines do not need
            res["also"] = also_block(args, lens, device)
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(lens, points_all, pupil_last)
        emit_line(res)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
