#!/usr/bin/env python3
"""bench.py -- rays/sec and PSFs/sec of the dual-pixel ray-traced PSF path.

Workload (BASELINE.json configs[1]): rf50mm refocused to 1 m (F/4 stop),
32x32x16 (x, y, z) PSF volume = 16384 point sources, 4096 primary rays
per point, 65x65 LEFT and RIGHT PSFs, lambda = 0.589 um.  One "step" = one
psf call over the rank's points (sdirt_amd.volume.VolumeStepper: ONE library call): draw the pupil uniforms
(torch CPU generator, the reference's order), upload, pupil mapping, chief-ray centre pass (2048 rays
per point) and the sample->trace->splat->normalise pass in ONE fused kernel launch, the reference's batch-global
Newton trip rule evaluated on the device (checked by the host `depth` steps later).

N > 1 is STRONG scaling by default (north_star): the ONE volume of the workload cut into N contiguous
shards -- config 2 on 8 GPUs: 2048 points per GPU --, every rank draws the same pupil samples, the trip masks are
OR-reduced over the ranks (one small all-reduce per step), and the [N/world, 2, ks, ks] blocks the kernels wrote are
all-gathered to every rank with ONE RCCL collective per step (485 MB received per rank and step at N = 8) on a side
stream under the next step's kernel -- `value` includes it, `value_no_gather` is the same loop without it; `gather`
carries the bytes a rank receives per step, the all-gather's own time on its stream and `gather_bound` (does it take
longer than the kernel it hides under?) and `volume_checksums_equal` (every rank's copy of the volume holds every rank's
shard bit for bit).  The run ends with `gather.trial`: the same K timed steps with the shards exchanged as seven concurrent
peer transfers per rank ('direct') instead of RCCL's all-gather, under a deadline; the faster one is `value`, both are
reported (`value_allgather`, `value_direct`; SDIRT_GATHER_TRIAL=0 or a set SDIRT_GATHER_ALGO skips it).  `--scaling weak`:
every rank renders its own 16384-point slab of a 32x32x(16*N) volume.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling strong|weak]   (N > 1: starts its own N ranks)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (see the driver contract) of less than 4 KB: the contract's keys, `roofline` (the
dominant kernel, k_psf_lr; the figures that say something -- valu_flops_frac, valu_issue_frac -- as flat scalars),
`cpu_baseline` (oracle/: a C port of the reference's CPU path, OpenMP over the host cores, and a PyTorch-CPU
restatement of the reference's whole-tensor execution model, on a bounded sample of the same workload; oracle/ is only
loaded for that leg), `also_summary` ([value, ms_per_step, roofline frac] of the staged SoA chain, f1, c4, c3, tcp, fit =
the loop of 1_fit_psfnet.py, c5 = config 5 end to end)
and `sweep_summary` (the strong-scaling compute side on this one GPU).  The FULL record -- per-kernel tables, counter
provenance, notes -- goes to the file named under `detail` (--detail-file, default bench_detail.json) and, with
--verbose, to stderr.
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
EXTRA_WORKLOADS = ("f1", "tcp", "staged", "sweep", "c5", "fit")


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


def _r(x, digits=6):
    """Floats of the one JSON line to six significant digits (the detail file keeps everything)."""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}") if math.isfinite(x) else None
    if isinstance(x, dict):
        return {k: _r(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, digits) for v in x]
    return x


LINE_LIMIT = 4000      # bytes: the driver keeps the parsed line whole only when it is short (16 KB was cut)
_TOP = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "world_size", "backend", "psfs_per_sec", "render_streams", "kernel_ms", "value_no_gather",
        "ms_per_step_no_gather", "value_two_streams", "ms_per_step_two_streams", "ms_per_step_sustained", "value_sustained",
        "sustained_steps", "value_pcie_inclusive", "value_allgather", "value_direct")
_CFG = ("workload", "name", "points_per_gpu", "spp", "ks", "parallelism", "gather", "newton_trip_policy",
        "relaunches_in_timed_region", "miopen_find_mode", "miopen_find_seconds", "backend")


def compact(res):
    """The ONE line the driver parses, under LINE_LIMIT bytes: the contract's keys, `roofline` and `cpu_baseline` with the
    figures that say something as flat scalars, one [value, ms_per_step, roofline frac] triple per side workload
    (`also_summary`), one [ms_per_step, kernel_ms, host_us_per_step, efficiency] row per shard size (`sweep_summary`).
    Everything else -- per-kernel tables, counter provenance, notes -- is in the detail file named under `detail`."""
    out = {k: res[k] for k in _TOP if k in res}
    cfg = res.get("config") or {}
    out["config"] = {k: cfg[k] for k in _CFG if k in cfg}
    rf = res.get("roofline")
    if rf:
        o = {k: rf.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic") if k in rf}
        for k in ("traffic_stale", "kernel", "algorithmic_bytes_per_launch"):
            if rf.get(k) is not None:
                o[k] = rf[k]
        vf, vi, rk = rf.get("valu_flops") or {}, rf.get("valu_issue") or {}, rf.get("rocprof_kernel_trace") or {}
        if vf:
            o["valu_flops_frac"], o["valu_tflops"] = vf.get("frac"), vf.get("achieved")
        if vi:
            o["valu_issue_frac"] = vi.get("frac")
            o["issue_bound_frac"] = (vi.get("issue_bound") or {}).get("frac")
            o["counters_stale"] = vi.get("stale")
        if rk:
            o["rocprof_median_ms"], o["counters"] = rk.get("median_ms"), rk.get("file")
        out["roofline"] = o
    cb = res.get("cpu_baseline")
    if cb:
        o = {k: cb[k] for k in ("value", "unit", "cores", "kind", "sample", "cpu_model") if k in cb}
        if "torch" in cb:
            o["torch_value"], o["torch_threads"] = cb["torch"]["value"], cb["torch"]["threads"]
        out["cpu_baseline"] = o
    g = res.get("gather")
    if g:
        out["gather"] = {k: g[k] for k in ("algo", "backend", "world_size", "gb_received_per_rank_per_step",
                                           "collectives_per_step", "ms", "GBps_received_per_rank", "compute_ms",
                                           "gather_bound", "volume_checksums_equal") if k in g}
        if g.get("trial"):
            out["gather"]["trial"] = {k: v for k, v in g["trial"].items() if k != "what"}
    also = res.get("also")
    if also:
        summ = {}
        for name, r in also.items():
            if name == "shard_sweep":
                continue
            if "error" in r:
                summ[name] = r["error"][:60]
            elif name == "staged":
                for ks_name, c in (r.get("staged") or {}).items():
                    # [rays/s of the chain, ms per step, the lowest HBM fraction among its three HBM-bound kernels]
                    fr = [c["kernels"][k]["frac_of_hbm_peak"] for k in ("sample_rays", "propagate_to", "forward_integral")
                          if "frac_of_hbm_peak" in c.get("kernels", {}).get(k, {})]
                    summ[f"staged_{ks_name}"] = [c.get("rays_per_s"), c.get("ms_per_step"), min(fr) if fr else None]
            else:
                summ[name] = [r.get("value"), r.get("ms_per_step"), (r.get("roofline") or {}).get("frac")]
        out["also_summary"] = summ
    sweep = res.get("shard_sweep") or (also or {}).get("shard_sweep", {}).get("shard_sweep")
    if sweep:
        out["sweep_summary"] = {name: [r.get("ms_per_step"), r.get("kernel_ms"), r.get("host_us_per_step"), r.get("efficiency")]
                                for name, r in sweep.items()}
        out["sweep_columns"] = ["ms_per_step", "kernel_ms", "host_us_per_step", "efficiency"]
    for k in ("kernels_ms",):
        if k in res and len(json.dumps(res[k])) < 200:
            out[k] = res[k]
    hg = res.get("hip_graph") or ((also or {}).get("c5") or {}).get("hip_graph")
    if hg:
        out["c5_ms_per_step_hip_graph" if also else "ms_per_step_hip_graph"] = hg.get("ms_per_step") or str(hg.get("error"))[:60]
    if res.get("detail"):
        out["detail"] = res["detail"]
    out = _r(out)
    # never longer than the limit: drop the optional blocks, largest first
    for k in ("sweep_summary", "also_summary", "kernels_ms", "gather"):
        if len(json.dumps(out)) <= LINE_LIMIT:
            break
        out.pop(k, None)
        out.pop("sweep_columns", None) if k == "sweep_summary" else None
    return out


DETAIL_PATH = None     # --detail-file: where the full record of the run goes (default bench_detail.json beside bench.py)
_EMIT = {"lock": None, "done": False}      # ONE line, whoever prints it (the main thread, or the gather trial's deadline thread)


def emit_line(res):
    """Rank 0's result: the FULL record into the detail file (and onto stderr under --verbose), its compact() form as the
    ONE JSON line on stdout."""
    import threading
    if _EMIT["lock"] is None:
        _EMIT["lock"] = threading.Lock()
    with _EMIT["lock"]:
        if _EMIT["done"]:
            return
        _EMIT["done"] = True
    path = DETAIL_PATH or os.path.join(ROOT, "bench_detail.json")
    try:
        with open(path, "w") as f:
            json.dump(res, f, indent=1)
        res = dict(res, detail=os.path.relpath(path, ROOT) if path.startswith(ROOT) else path)
    except OSError as e:          # a read-only checkout must not cost the line
        sys.stderr.write(f"bench.py: detail file not written: {e}\n")
    if os.environ.get("SDIRT_BENCH_VERBOSE") == "1":
        sys.stderr.write(json.dumps(res) + "\n")
    line = (json.dumps(compact(res)) + "\n").encode()
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
    16384 / 8192 / 4096 / 2048 points per step (what a rank renders at world 1 / 2 / 4 / 8 under strong scaling), each
    through the loop `--gpus N` runs (sdirt_amd.volume.VolumeStepper with force_collectives): one library call per step,
    the mask all-reduce in front of the device-side trip rule, the shard rendered into its [n, 2, ks, ks] block and that
    block all-gathered on the comm stream -- every collective really issued, on a world-1 RCCL process group.  What a
    1-GPU box cannot show is the xGMI side: a rank of an 8-GPU run RECEIVES 7 blocks where this stand-in copies its own.
    Every k-th point of the volume (all depth planes, all field positions), so that the batch-global Newton trip
    tables are the whole volume's.

    Every figure is physical: ms_per_step = wall time between two fences (barrier + device synchronisation on both
    sides, pipeline drain included) / steps, over >= 500 steps for shards of <= 4096 points; efficiency = (ms_per_step of
    the plain 16384-point single-GPU loop x n / 16384) / ms_per_step <= ~1: a rank of an N-GPU run would take exactly 1/N
    of the single-GPU step at 1.0.  kernel_ms = HIP events around the step's library call (48 KB upload + pupil mapping
    + fused kernel); with two render streams consecutive launches overlap and a pair spans both."""
    import socket
    import torch.distributed as dist
    from sdirt_amd.volume import VolumeStepper
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
    try:
        if lens is None:
            lens = build_lens(dev)
        pts_all = volume_points(1, "c2").to(dev)
        n_full = pts_all.shape[0]
        ks, spp = WORKLOADS["c2"]["ks"], WORKLOADS["c2"]["spp"]
        rows = {}
        import gc

        def run(points, force, label, streams=1):
            n = points.shape[0]
            steps = max(args.steps, 100 if n > 4096 else 500)
            torch.manual_seed(0)
            st = VolumeStepper(lens, points, n, ks, spp, DP, gather=force, streams=streams, time_steps=True,
                               force_collectives=force)
            for _ in range(20):
                st.step()
            st.fence()
            gc.collect()
            gc.freeze()
            st.reset_counters()
            r0 = st.relaunches
            dt = st.timed(steps)
            row = {"points_per_step": n, "steps": steps, "ms_per_step": dt / steps * 1e3, "kernel_ms": st.kernel_ms(),
                   # wall time the host spends in step() and settle() per step, without the time it sits blocked waiting
                   # for a control block: draw, ONE library call, the collectives' host side, event bookkeeping
                   "host_us_per_step": st.t_step / steps * 1e6, "host_blocked_us_per_step": st.t_wait / steps * 1e6,
                   "rays_per_s": n * spp * steps / dt, "relaunches_in_timed_region": st.relaunches - r0,
                   "render_streams": streams, "steps_in_flight": st.depth,
                   "trip_tables": [[int(v) for v in t_] for t_ in st.tables]}
            g_ms = st.gather_ms()
            if g_ms is not None:
                row["gather_ms"], row["gather_block_mb"] = g_ms, n * 2 * ks * ks * 4 / 1e6
            rows[label] = row
            del st
            torch.cuda.empty_cache()
            return row
        base = run(pts_all, False, "single_gpu_loop_16384")       # the headline's own loop: no process group in the way
        base2 = run(pts_all, False, "single_gpu_loop_16384_two_streams", 2)

        def rate(row):
            # against the single-GPU loop with the SAME number of render streams: a rank with two streams is compared with
            # a single GPU that also overlaps its launches
            ref = base if row["render_streams"] == 1 else base2
            row["efficiency"] = ref["ms_per_step"] * (row["points_per_step"] / n_full) / row["ms_per_step"]
            if row["render_streams"] == 1:
                row["kernel_efficiency"] = base["kernel_ms"] * (row["points_per_step"] / n_full) / row["kernel_ms"]
                row["gpu_idle_us_per_step"] = max(0.0, (row["ms_per_step"] - row["kernel_ms"]) * 1e3)
            row["trip_tables_equal_full_batch"] = row.pop("trip_tables") == base["trip_tables"]
        for k in (1, 2, 4, 8):
            for streams in (1, 2):
                row = run(pts_all[::k].contiguous(), True, f"world{k}_shard_{n_full // k}" + ("" if streams == 1 else "_two_streams"), streams)
                row["as_rank_of_world"] = k
                rate(row)
        base["efficiency"] = base2["efficiency"] = 1.0
        base2["speedup_over_one_stream"] = base["ms_per_step"] / base2["ms_per_step"]
        base.pop("trip_tables", None)
        base2.pop("trip_tables", None)
    finally:
        if own_group:
            dist.destroy_process_group()
    last = rows[f"world8_shard_{n_full // 8}"]
    res = {"metric": "rays/sec rf50mm 65x65 DP-PSF @4096spp, strong-scaling compute side on one GPU (2048-point step of a rank of 8)",
           "value": last["rays_per_s"], "unit": "rays/s", "n_gpus": 1, "steps": last["steps"], "warmup": 20,
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
    The depth network's convolutions are stock MIOpen, run as a user of the network runs them: with
    torch.backends.cudnn.benchmark = True, i.e. ONE find pass over the ~30 forward shapes during the first frame
    (`miopen_find_seconds`; torch keeps the pick per shape for the life of the process, so the flag must be set before
    the first call).  SDIRT_C5_FIND=0: MIOpen's immediate-mode picks instead -- the fallback solvers VERDICT r05 found in
    the driver's stderr (GemmFwdRest without its workspace): 7.1 instead of ~3.5 ms for the forward pass.
    `roofline`: the frame's largest kernel of this package, k_psfnet_mlp (MFMA-bound; its own HIP-event time)."""
    from sdirt_amd.dfdp import DfDPNet
    from sdirt_amd.psfnet import PSFNet
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback exists)"
    dev = torch.device("cuda", 0)
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

    def chain(ev=None):
        with torch.no_grad():
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if ev is not None else None
            if e:
                e[0].record(stream)
            pair = m.render(img, depth, foc)
            if e:
                e[1].record(stream)
            left, right = pair[:, :3].contiguous(), pair[:, 3:].contiguous()
            with torch.autocast("cuda", dtype=torch.float16):
                disp = net(left, right)
            if e:
                e[2].record(stream)
                ev["render"].append((e[0], e[1]))
                ev["dfdp_forward"].append((e[1], e[2]))
        return disp

    def measure(warm):
        t0 = time.perf_counter()
        for _ in range(warm):
            disp = chain()
        torch.cuda.synchronize(dev)
        warm_s = time.perf_counter() - t0
        steps = max(args.steps, 5)
        ev = {"render": [], "dfdp_forward": []}
        t0 = time.perf_counter()
        for _ in range(steps):
            disp = chain(ev)
        torch.cuda.synchronize(dev)
        wall = (time.perf_counter() - t0) / steps * 1e3
        assert bool(torch.isfinite(disp).all())
        return wall, {k: float(np.mean([a.elapsed_time(b) for a, b in v])) for k, v in ev.items()}, warm_s, steps, disp
    was = bool(torch.backends.cudnn.benchmark)
    find = os.environ.get("SDIRT_C5_FIND", "1") == "1"
    torch.backends.cudnn.benchmark = find
    wall, ms, first_s, steps, disp = measure(max(args.warmup, 3))
    # the same frame captured once in a hipGraph and replayed (sdirt_amd.graphs.GraphedCall): same kernels, same arguments,
    # no hand-over gaps between the ~85 dependent launches.  A side figure: a capture that fails costs only itself.
    graph_ms = graph_err = None
    try:
        from sdirt_amd.graphs import GraphedCall
        frame = GraphedCall(chain, warmup=2, device=dev)
        for _ in range(3):
            frame()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            g_disp = frame()
        torch.cuda.synchronize(dev)
        graph_ms = (time.perf_counter() - t0) / steps * 1e3
        graph_diff = float((g_disp.float() - disp.float()).abs().max())
    except Exception as e:                  # noqa: BLE001
        graph_err = f"{type(e).__name__}: {str(e)[:200]}"
    torch.backends.cudnn.benchmark = was
    # the PSF network alone (both passes of PSFNet.pred): 786432 rows x 4.78 MFLOP, the largest kernel of this package
    # in the frame
    x, y = torch.meshgrid(torch.linspace(-1, 1, W), torch.linspace(1, -1, H), indexing="xy")
    o = torch.stack((x.to(dev)[None], y.to(dev)[None], m.depth2z(depth + m.d_sensor).squeeze(1)), -1).float()
    with torch.no_grad():
        _, mlp_ms = _hip_ms(lambda: m.psfnet.forward_fused(o, mirror=True), 10, 3, dev)
    macs = 3 * 128 + 128 * 512 + 8 * 512 * 512 + 512 * ks * ks
    flops = 2 * macs * 2 * H * W
    tf = flops / (mlp_ms * 1e-3) / 1e12
    MFMA_PEAK_F16 = 2500.0       # MI355X_MICROARCH.md: dense fp16 / bf16 matrix peak, TFLOP/s
    res = {"metric": "frames/sec config 5: RGB-D 512x768 -> PSFNet.render (DP pair) -> DfDP net forward, 1 GPU",
           "value": 1e3 / wall, "unit": "frames/s", "n_gpus": 1, "steps": steps, "warmup": max(args.warmup, 3),
           "ms_per_step": wall, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16",
           "data": "synthetic",
           "config": {"workload": "BASELINE config 5: synthetic RGB-D frame 1x3x512x768 (depth U[0.5, 5] m, focus 1 m) -> PSFNet.render, ks 21, "
                                  "full-size PSF network (seeded weights) -> DfDPNet forward (fp16 autocast), rf50mm", "name": "c5",
                      "miopen_find_mode": bool(find), "miopen_find_seconds": first_s if find else None, "first_calls_s": first_s},
           "kernels_ms": dict(ms, psfnet_mlp=mlp_ms),
           "hip_graph": ({"ms_per_step": graph_ms, "value": 1e3 / graph_ms, "max_abs_diff_vs_eager": graph_diff,
                          "what": "the same frame captured once (sdirt_amd.graphs.GraphedCall) and replayed, wall time over the same number of frames"}
                         if graph_ms is not None else {"error": graph_err}),
           "roofline": {"bound": "mfma", "achieved": tf, "peak": MFMA_PEAK_F16, "unit": "TFLOP/s", "frac": tf / MFMA_PEAK_F16,
                        "traffic": None, "kernel": "k_psfnet_mlp (3 -> 128 -> 512 x 9 -> 441, both passes of PSFNet.pred: 786432 rows)",
                        "algorithmic_flops_per_launch": flops,
                        "note": "the frame's largest kernel of this package (2.9 of the render's 3.4 ms); the depth network's "
                                "convolutions are stock MIOpen"}}
    if emit:
        emit_line(res)
    return res


def bench_fit(args, emit=True):
    """The loop of 1_fit_psfnet.py at its own settings (PSFNet.train_psfnet, psfnet.py:101-168; 1_fit_psfnet.py:36: bs 64, spp
    20000, ks 21, the full-size MLP): every iteration ray-traces a fresh batch of 64 PSFs (64 x 20000 primary + 64 x 2048
    chief-ray rays through the fused kernel, trip tables verified on the device) and takes one AdamW step on the PSF
    network under fp16 autocast with loss scaling -- forward + backward replayed from a hipGraph, batches ray-traced two
    ahead on a second stream (train_psfnet(pipelined=True), the default on the GPU).  A step = one iteration; `value` =
    iterations per second over max(steps, 1000) iterations of ONE train_psfnet call, its set-up (optimiser, graph capture:
    ~0.1 s) included, after a 20-iteration call that has found the trip tables."""
    import tempfile
    from sdirt_amd.psfnet import PSFNet
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback exists)"
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    np.random.seed(0)
    m = PSFNet(os.path.join(ROOT, "sdirt_amd", "data", "rf50mm.json"), sensor_res=(512, 768), kernel_size=21, device=dev)
    m.refocus(-1000 + m.d_sensor)
    iters = max(args.steps, 1000)
    with tempfile.TemporaryDirectory() as tmp:
        kw = dict(bs=64, lr=1e-4, spp=20000, evaluate_every=10 ** 9, result_dir=tmp, figures=False)
        m.train_psfnet(iters=20, **kw)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        losses = m.train_psfnet(iters=iters - 1, **kw)               # (the loop runs iters + 1 steps, as the reference's)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
    n = len(losses)
    res = {"metric": "iterations/sec of 1_fit_psfnet.py's loop (64 ray-traced 21x21 PSFs @20000spp + one AdamW step of the PSF network)",
           "value": n / dt, "unit": "iterations/s", "n_gpus": 1, "steps": n, "warmup": 20, "ms_per_step": dt / n * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32 rays / f16 autocast network", "data": "synthetic",
           "rays_per_sec": n * 64 * 20000 / dt,
           "config": {"workload": "1_fit_psfnet.py:36 train_psfnet(bs=64, spp=20000), rf50mm, ks 21, MLP 3-128-512x9-441, AdamW + cosine "
                                  "schedule, fp16 autocast + loss scaling; PSF batches from the fused HIP kernel", "name": "fit",
                      "host_relaunches_total": int(m.trips.relaunches), "device_corrections_total": int(m.trips.device_relaunches)},
           "loss_first_last": [float(losses[0]), float(losses[-1])]}
    if emit:
        emit_line(res)
    return res


def quick_volume(workload, steps, device):
    """K steps of another PSF-volume workload (WORKLOADS) through the headline's loop (VolumeStepper) -- for the `also`
    block of the default line."""
    from sdirt_amd.volume import VolumeStepper
    wl = WORKLOADS[workload]
    lens = build_lens(device, wl["lens"], wl["sensor_z"])
    gz = 8 if workload.startswith("c3") else wl["grid_z"]
    pts = volume_points(1, workload).to(device)
    n, ks, spp = pts.shape[0], wl["ks"], wl["spp"]
    st = VolumeStepper(lens, pts, n, ks, spp, DP, time_steps=True)
    for _ in range(3):
        st.step()
    st.fence()
    r0 = st.relaunches
    dt = st.timed(steps)
    k_ms = st.kernel_ms()
    alg_bytes = n * (12 + 8) + (spp + 2048) * 8 + 2 * n * ks * ks * 4
    ach = alg_bytes / (k_ms * 1e-3) / 1e9
    return {"metric": f"rays/sec {wl['lens']} {ks}x{ks} DP-PSF @{spp}spp", "value": n * spp * steps / dt, "unit": "rays/s",
            "steps": steps, "ms_per_step": dt / steps * 1e3, "kernels_ms": {"psf_call": k_ms},
            "roofline": {"bound": "valu", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "traffic": None, "algorithmic_bytes_per_launch": alg_bytes},
            "config": {"workload": wl["desc"].format(gz=gz) + f", {n} points, {spp} spp, {ks}x{ks} L+R", "name": workload,
                       "relaunches_in_timed_region": st.relaunches - r0}}


def also_block(args, lens, device):
    """The other driver-timed lines of the default run: the staged SoA chain (HBM-bound kernels; call by call and through
    the fused entries), the per-pixel PSF convolution f1 (HBM-bound), config 4 (rf35mm, the second prescription), one
    GPU's share of config 3 (8192 points x 8192 spp, 21 x 21), the reference's own timing harness (tcp), the loop of
    1_fit_psfnet.py (fit: 1000 iterations), config 5 end to end (c5) -- and the strong-scaling compute side (shard_sweep)."""
    import copy
    q = copy.copy(args)
    q.steps, q.warmup, q.sustain_seconds = 20, 2, 0.0
    out = {}
    # the staged chain's steps are 3 ms: warmed for ~0.1 s first -- the clock of a chip that has just been idle (set-up,
    # allocations) dips for some tens of ms shortly after work resumes (tools/clock_ramp.py), longer than five such steps
    qs = copy.copy(q)
    qs.steps, qs.warmup = 10, 40
    qs.staged_chain = "both"           # the call-by-call chain and the one through the fused entries (`fused_calls`)
    # (fit first: its loop overlaps two streams -- PSF batches beside the step's graph -- and HIP hands hardware queues to streams
    # in order of creation: behind the other workloads' streams its two ended up sharing a queue, 1.17 instead of 0.84 ms per iteration)
    for name, fn in (("fit", lambda: bench_fit(q, emit=False)),
                     ("staged", lambda: bench_staged(qs, emit=False, lens=lens)),
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
                    "value_pcie_inclusive", "reference_harness", "shard_sweep", "hip_graph")
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


TRIAL_DEADLINE_S = float(os.environ.get("SDIRT_BENCH_TRIAL_DEADLINE_S", "120"))


def gather_algo_trial(loop, steps, dt_default, res, rays):
    """N > 1, LAST thing of the run (every rank calls it; `res` is rank 0's finished record, None elsewhere): the same K
    timed steps once more with the shards exchanged as seven concurrent peer transfers per rank (dist.all_gather_shards
    algo 'direct') instead of RCCL's all-gather -- xGMI is a full mesh of point-to-point links, and which of the two
    feeds a rank's seven links better is a question for the node.  Measured like the
    headline (VolumeStepper.timed: K steps between two fences, MAX over ranks, so every rank sees the same figures and
    takes the same decision), checksums compared as for the headline; on its own communicator, so nothing of the
    product's is touched.  The faster of the two becomes `value` (a margin of 3 %), both stay in `gather.trial`.

    An experiment must not cost the record: it runs under a deadline.  If it has not returned by then, or raises,
    rank 0 prints the record it already has (RCCL's all-gather, complete) and every rank leaves at once -- there is no
    orderly way out of a collective that a peer never entered."""
    import threading
    import torch.distributed as dist

    def leave(why):
        if res is not None:
            res["gather"]["trial"] = {"allgather_ms_per_step": dt_default / steps * 1e3, "direct": why, "adopted": "allgather"}
            emit_line(res)
        sys.stderr.flush()
        os._exit(0)
    done = threading.Event()
    watch = threading.Thread(target=lambda: done.wait(TRIAL_DEADLINE_S) or leave(f"no result within {TRIAL_DEADLINE_S:.0f} s"), daemon=True)
    watch.start()
    keep = (loop.gather_group, loop.gather_algo)
    try:
        if os.environ.get("SDIRT_BENCH_FAKE_TRIAL") == "hang":       # (tests: the deadline path)
            time.sleep(10 * TRIAL_DEADLINE_S)
        if os.environ.get("SDIRT_BENCH_FAKE_TRIAL") == "raise":
            raise RuntimeError("SDIRT_BENCH_FAKE_TRIAL")
        loop.gather_group, loop.gather_algo = dist.new_group(), "direct"
        for _ in range(3):                 # the peer connections come up with the first transfers
            loop.step()
        loop.fence()
        loop.gather_ms()                   # (forget the events of the warm-up)
        dt_direct = loop.timed(steps)
        ok = loop.verify_gather()
        g_ms = loop.gather_ms()            # the exchange's own time on the comm stream, MAX over ranks like the headline's
        if g_ms is not None:
            gm = torch.tensor([g_ms], dtype=torch.float64, device=loop.device)
            dist.all_reduce(gm, op=dist.ReduceOp.MAX)
            g_ms = float(gm.item())
    except Exception as e:                 # noqa: BLE001 -- whatever it is, the record comes first
        leave(f"{type(e).__name__}: {str(e)[:200]}")
    finally:
        done.set()
    loop.gather_group, loop.gather_algo = keep
    if res is not None:
        adopt = bool(ok) and dt_direct < 0.97 * dt_default
        res["gather"]["trial"] = {"allgather_ms_per_step": dt_default / steps * 1e3, "direct_ms_per_step": dt_direct / steps * 1e3,
                                  "direct_volume_checksums_equal": ok, "direct_gather_ms": g_ms, "adopted": "direct" if adopt else "allgather",
                                  "what": "the K timed steps once more with algo 'direct' (one send + one receive per peer in one "
                                          "group), same fences, MAX over ranks; adopted as `value` when checksums hold and it "
                                          "is more than 3 % faster"}
        res["value_allgather"], res["value_direct"] = rays / dt_default, rays / dt_direct
        if adopt:
            res["value"], res["ms_per_step"] = rays / dt_direct, dt_direct / steps * 1e3
            res["psfs_per_sec"] = res["psfs_per_sec"] * dt_default / dt_direct
            g = res["gather"]
            g["algo"], g["volume_checksums_equal"] = "direct", ok
            if g_ms:
                g["ms"], g["GBps_received_per_rank"] = g_ms, g["gb_received_per_rank_per_step"] / (g_ms * 1e-3)
                g["gather_bound"] = bool(g_ms > g["compute_ms"])


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
    ap.add_argument("--scaling", choices=("weak", "strong"), default="strong",
                    help="N > 1: strong (default; SURVEY.md §8e, north_star) = the ONE volume of the workload (c2 / c4: 16384 "
                         "points, c3: 65536) cut into N contiguous shards -- config 2 on 8 GPUs is 2048 points per GPU and "
                         "485 MB received per rank and step; weak = every rank renders its own 16384-point slab of a volume "
                         "N times as deep (a 131072-point volume at N = 8, 3.9 GB received per rank and step)")
    ap.add_argument("--detail-file", default=None,
                    help="where the full record of the run goes (default: bench_detail.json beside bench.py); the ONE "
                         "line on stdout is its compact form (< 4 KB)")
    ap.add_argument("--verbose", action="store_true", help="also print the full record on stderr")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-two-streams", action="store_true",
                    help="N = 1: skip the second leg (the same K steps on two alternating render streams) -- the profiling "
                         "recipes: overlapping launches would stand in the kernel trace with twice their duration")
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
                         "group; c5: config 5 end to end (RGB-D frame -> PSFNet.render -> DfDP net forward); fit: the loop of "
                         "1_fit_psfnet.py (a ray-traced batch of 64 PSFs @20000 spp + one AdamW step per iteration)")
    ap.add_argument("--staged-ks", default="65,21", help="--workload staged: the grid sizes to run the chain for")
    ap.add_argument("--staged-chain", choices=("both", "calls"), default="both",
                    help="--workload staged: `calls` times the call-by-call chain only (the profiling recipe: one kernel "
                         "per name), `both` also the chain through sdirt_trace2sensor / SDIRT_PSF_NORMALIZE")
    args = ap.parse_args()
    global DETAIL_PATH
    DETAIL_PATH = args.detail_file
    if args.verbose:
        os.environ["SDIRT_BENCH_VERBOSE"] = "1"
    if not (args.gpus > 1 and "WORLD_SIZE" not in os.environ):      # (the self-launching parent relays its children's output)
        claim_stdout()
    if args.workload in EXTRA_WORKLOADS:
        assert args.gpus == 1, f"--workload {args.workload} is a single-GPU measurement"
        return {"f1": bench_f1, "tcp": bench_tcp, "staged": bench_staged, "sweep": bench_sweep, "c5": bench_c5,
                "fit": bench_fit}[args.workload](args)
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
    from sdirt_amd.volume import VolumeStepper
    lens = build_lens(device, wl["lens"], wl["sensor_z"])
    strong = args.scaling == "strong"
    # strong (default): the workload's own volume (c3 is stated for a node of 8: 64 planes) whatever the number of ranks;
    # weak: the volume grows with the ranks (16 -- c3: 8 -- depth planes per GPU).  One GPU: the same thing.
    points_all = volume_points((8 if args.workload.startswith("c3") and world > 1 else 1) if strong else world, args.workload)
    n_total = points_all.shape[0]
    a, b = sd.shard_bounds(n_total, world)[rank]
    points_local = points_all[a:b].to(device)
    n_local = b - a
    gather_default = world > 1 and not args.no_gather
    # N > 1: consecutive steps alternate between two render streams (the next step's workgroups fill what is left of
    # the previous launch's end).  N = 1 keeps ONE stream: the headline's kernel time is then the time of a launch that has
    # the chip to itself, which is what `roofline` and the committed rocprofv3 summaries are about.
    n_streams = 2 if world > 1 else 1
    # every rank draws the same pupil uniforms from its own generator (VolumeStepper checks that they do)
    torch.manual_seed(0)
    # one-off initialisation, like loading the library: the stepper's first call on a lens discovers the Newton trip
    # tables (a launch with 10 trips everywhere, then the verified table); they are lens state from then on
    loop = VolumeStepper(lens, points_local, n_total, KS, SPP, DP, gather=gather_default, streams=n_streams, time_steps=True)
    # Everything the run needs is built: park the interpreter's heap in the permanent generation.  A full collection of
    # torch's heap takes 40-55 ms (measured: it strikes ~150 ms after the first launch, i.e. inside the timed window of
    # a --warmup 5 --steps 20 run, idles the GPU and costs it its clock: tools/clock_ramp.py); after the freeze the
    # collector only ever walks what the steps themselves allocate.
    import gc
    gc.collect()
    gc.freeze()
    for _ in range(args.warmup):
        loop.step()
    loop.fence()
    loop.reset_counters()
    relaunch0 = loop.relaunches
    dt = loop.timed(args.steps)
    relaunches = loop.relaunches - relaunch0
    k_event_ms = loop.kernel_ms()
    # N > 1: every rank's copy of the volume holds every rank's shard, bit for bit (checksums of the last step's blocks)
    gather_ok = loop.verify_gather() if gather_default else None
    host_us = loop.t_step / args.steps * 1e6
    # the all-gathers of the timed steps as the comm stream saw them (start = the shard is final and the previous
    # gather has left the stream, end = every rank's shard has arrived here); MAX over ranks like the wall time
    gather_ms = loop.gather_ms() if gather_default else None
    if gather_ms is not None:
        gm = torch.tensor([gather_ms], dtype=torch.float64, device=device)
        dist.all_reduce(gm, op=dist.ReduceOp.MAX)
        gather_ms = float(gm.item())

    # the same K steps without the all-gather (N > 1), and a loop long enough for the clocks to
    # settle (the K-step region lasts a fraction of a second)
    dt_ng = None
    if gather_default:
        loop.gather = False
        dt_ng = loop.timed(args.steps)
        loop.gather = True
    k_sus = dt_sus = None
    if args.sustain_seconds > 0:
        k_sus = max(args.steps, int(math.ceil(args.sustain_seconds / (dt / args.steps))))
        dt_sus = loop.timed(k_sus)
    pupil_last = None
    if world == 1 and not args.no_cpu_baseline:
        # the pupil points of the GPU leg's last step, as the device mapped them (they stand in the slot's scratch)
        s_ = loop._slots[(loop.steps - 1) % len(loop._slots)]
        xy = s_.scratch[loop.scratch_bytes - 4 * loop.n_u:].view(torch.float32).clone()
        pupil_last = (xy[:SPP], xy[SPP:2 * SPP], xy[2 * SPP:2 * SPP + 2048], xy[2 * SPP + 2048:])
    tables = [[int(v) for v in t_] for t_ in loop.tables]

    # N = 1: the same K steps once more with consecutive steps on two alternating streams (what a pipelined consumer of
    # batch after batch gets: the next launch's workgroups fill what is left of the previous launch's end)
    dt_two = None
    if world == 1 and not args.no_two_streams:
        del loop
        torch.cuda.empty_cache()
        loop2 = VolumeStepper(lens, points_local, n_total, KS, SPP, DP, streams=2)
        for _ in range(max(args.warmup, 2)):
            loop2.step()
        dt_two = loop2.timed(args.steps)
        del loop2
        torch.cuda.empty_cache()

    if rank == 0:
        rays = n_total * SPP * args.steps
        # algorithmic HBM bytes of ONE k_psf_lr launch: read the points,
        # centres and pupil samples once, write the L and R tiles once.
        alg_bytes = n_local * (12 + 8) + (SPP + 2048) * 8 + 2 * n_local * KS * KS * 4
        dom = "psf_call"
        k_ms = {dom: k_event_ms}
        if n_streams > 1:
            # two launches overlap on the chip: an event pair spans both.  What a launch costs the step is the step itself
            # (without the gather): that is the duration the roofline figures and `gather_bound` are computed with.
            k_ms[dom] = min(k_event_ms, (dt_ng if dt_ng is not None else dt) / args.steps * 1e3)
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
        cfg_name = {"c2": "BASELINE config 2", "c3": "BASELINE config 3", "c3k65": "BASELINE config 3 on 65x65 grids",
                    "c4": "BASELINE config 4"}[args.workload]
        gz_total = (64 if args.workload.startswith("c3") and world > 1 else GRID_Z) if strong else GRID_Z * world
        res = {
            "metric": f"rays/sec {wl['lens']} {KS}x{KS} DP-PSF @{SPP}spp", "value": rays / dt,
            "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "world_size": dist.get_world_size() if world > 1 else 1,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "psfs_per_sec": n_total * args.steps / dt,
            "backend": backend,
            "config": {"workload": f"{cfg_name}: " + wl["desc"].format(gz=gz_total)
                                   + (f" = {n_total} points, ONE volume cut into {world} contiguous shards" if strong and world > 1 else
                                      (f" = {n_total} points (weak scaling: a slab of {n_local} per GPU)" if world > 1 else f" = {n_total} points"))
                                   + f", {n_local} points/GPU, {SPP} spp (+2048 chief-ray "
                                   f"rays/point), {KS}x{KS} L+R PSFs, lambda 0.589um, focus 1 m F/4",
                       "name": args.workload,
                       "points_per_gpu": n_local, "points_total": n_total, "spp": SPP, "ks": KS,
                       "parallelism": f"points sharded over {world} GPU(s)"
                                      + ("" if world == 1 else
                                         ", every rank draws the same pupil samples, batch-global "
                                         "Newton trip check (mask all-reduce) per step")
                                      + (" + ONE RCCL all-gather of the [N/world, 2, ks, ks] blocks to every rank"
                                         if gather_default else ""),
                       "gather": bool(gather_default),
                       "newton_trip_policy": lens.trip_policy,
                       "trip_tables": tables,
                       "relaunches_in_timed_region": relaunches},
            "kernels_ms": k_ms, "render_streams": n_streams, "kernel_ms": k_ms[dom],
            "kernel_event_ms": k_event_ms, "host_us_per_step": host_us,
            "kernel_ms_what": "HIP events on the render stream around the step's ONE library call (sdirt_psf_call: 48 KB upload, "
                              "one pupil-mapping launch, k_psf_lr); the committed rocprofv3 kernel trace is k_psf_lr alone",
            # what bounds the kernel is its vector ALU time (profiles/r03/k_psf_lr_sites.txt):
            # `valu_flops` is the fraction that says something; achieved / peak / frac / traffic are the
            # mandated HBM figures (0.7 % by construction: 8 algorithmic bytes against 3.4 kflop per ray)
            "roofline": {"bound": "valu", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_stale": bool(counters["stale"]) if counters else None,
                         "valu_flops": valu_flops,
                         "kernel": "k_psf_lr<R,small-r,Lean,CENTER> (chief-ray + primary pass of a point in one workgroup)",
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
                                    if valu else "~5 k", alg_bytes / max(1, n_local * (SPP + 2048)))},
        }
        if dt_ng is not None:
            res["value_no_gather"] = rays / dt_ng
            res["ms_per_step_no_gather"] = dt_ng / args.steps * 1e3
            gb = 2 * (n_total - n_local) * KS * KS * 4 / 1e9      # received per rank and step
            compute_ms = k_ms[dom]
            width = max(b_ - a_ for a_, b_ in sd.shard_bounds(n_total, world))
            res["gather"] = {"algo": os.environ.get("SDIRT_GATHER_ALGO", "allgather"), "volume_checksums_equal": gather_ok,
                             "backend": dist.get_backend(), "world_size": dist.get_world_size(),
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
        if world == 1 and args.workload == "c2" and not args.no_also:
            res["also"] = also_block(args, lens, device)
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(lens, points_all, pupil_last)
    else:
        res = None
    if gather_default and "SDIRT_GATHER_ALGO" not in os.environ and os.environ.get("SDIRT_GATHER_TRIAL", "1") == "1":
        gather_algo_trial(loop, args.steps, dt, res, n_total * SPP * args.steps)
    if rank == 0:
        emit_line(res)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
