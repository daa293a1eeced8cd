#!/usr/bin/env python3
"""What stands between two fused kernels of consecutive steps of the PSF-volume render loop (VolumeStepper)?

  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d OUT -- python3 tools/step_timeline.py run [--n 2048] [--steps 300] [--collectives]
  python3 tools/step_timeline.py report OUT

`run` steps a 2048-point shard of config 2 (every 8th point) through the loop; `report` reads the trace: per pair of
consecutive k_psf_lr launches the idle time between them and the dispatches / copies that ran in it."""
import argparse
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(args):
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    import torch
    import bench
    from sdirt_amd.volume import VolumeStepper
    dev = torch.device("cuda", 0)
    if args.collectives:
        import torch.distributed as dist
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1, device_id=dev)
    lens = bench.build_lens(dev)
    pts = bench.volume_points(1, "c2")[:: 16384 // args.n].contiguous().to(dev)
    torch.manual_seed(0)
    st = VolumeStepper(lens, pts, pts.shape[0], 65, 4096, bench.DP, gather=args.collectives and not args.no_gather, force_collectives=args.collectives,
                       streams=args.streams, time_steps=True)
    for _ in range(20):
        st.step()
    st.fence()
    if args.freeze:
        import gc
        gc.collect()
        gc.freeze()          # a full collection of torch's heap takes 40-55 ms: longer than the ~20 ms of work the loop keeps queued
    dt = st.timed(args.steps)
    print(f"{pts.shape[0]} points, {args.steps} steps: {dt / args.steps * 1e3:.4f} ms per step, host {st.t_step / (args.steps + 20) * 1e6:.0f} us per step, "
          f"library call {st.kernel_ms():.4f} ms by HIP events, gather {st.gather_ms()} ms, {st.depth} steps in flight")


def report(out):
    rows = []
    for f in glob.glob(os.path.join(out, "**", "*_kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:48], "q" + r.get("Queue_Id", "?")))
    for f in glob.glob(os.path.join(out, "**", "*_memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "?"), "copy"))
    rows.sort()
    main = [r for r in rows if "k_psf_lr" in r[2]]
    main = main[len(main) // 2:]                       # the timed half
    gaps, between = [], {}
    for a, b in zip(main, main[1:]):
        gaps.append((b[0] - a[1]) / 1e3)
        for r in rows:
            if a[1] - 2000 <= r[0] < b[0] and r is not a and r is not b:
                k = (r[2], r[3])
                e = between.setdefault(k, [0, 0.0, 0.0])
                e[0] += 1; e[1] += (r[1] - r[0]) / 1e3; e[2] += (r[0] - a[1]) / 1e3
    import statistics
    dur = [(r[1] - r[0]) / 1e3 for r in main]
    print(f"{len(main)} k_psf_lr launches: duration median {statistics.median(dur):.1f} us; idle between consecutive launches: "
          f"median {statistics.median(gaps):.1f} us, mean {statistics.mean(gaps):.1f}, p90 {sorted(gaps)[int(0.9 * len(gaps))]:.1f}, max {max(gaps):.1f}")
    print("what ran between two launches (per gap: count, mean duration us, mean start offset after the previous kernel's end us):")
    for k, (n, d, o) in sorted(between.items(), key=lambda kv: -kv[1][0])[:14]:
        print(f"  {n / len(gaps):5.2f} x  {d / n:7.1f} us  @ {o / n:7.1f} us   {k[0]}  [{k[1]}]")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=("run", "report"))
    ap.add_argument("out", nargs="?")
    ap.add_argument("--n", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--streams", type=int, default=1)
    ap.add_argument("--collectives", action="store_true")
    ap.add_argument("--no-gather", action="store_true", help="--collectives without the all-gather (the mask all-reduce only)")
    ap.add_argument("--freeze", action="store_true", help="gc.freeze() before the timed steps (what bench.py does)")
    a = ap.parse_args()
    run(a) if a.mode == "run" else report(a.out)
