#!/bin/bash
# Kernel trace of BASELINE config 5 end to end (bench.py --workload c5: PSFNet.render -> DfDPNet forward), per-kernel table:
#   tools/profile_c5.sh [round-dir, default r06]   -> gpurun_out/prof_<round>_c5/kernel_stats_bench_c5.csv + top_kernels.txt
# The find pass of bench_c5 runs inside the trace (its search launches are in the table: the steady-state per-frame
# figures are the HIP-event times bench.py prints; the table says WHICH kernels MIOpen ends up running).
set -u
RND=${1:-r06}
OUT=gpurun_out/prof_${RND}_c5
mkdir -p "$OUT"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --workload c5 --steps 20 --detail-file "$OUT/bench_c5_detail.json" > "$OUT/bench_c5_line.json" 2> "$OUT/trace.log"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/trace/**/*_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
# the last 20 FRAMES of the timed loop: a frame starts with PSFNet.render's degamma (k_tone<0>); the run ends with ten
# launches of the PSF network alone (bench_c5's own timing of k_psfnet_mlp), which are not part of a frame
starts = [int(r["Start_Timestamp"]) for r in rows if "k_tone<0>" in r["Kernel_Name"] or "k_toneILi0E" in r["Kernel_Name"]]
assert len(starts) >= 21, len(starts)
lo, hi = starts[-21], starts[-1]
agg = collections.defaultdict(lambda: [0, 0.0])
tail = [r for r in rows if lo <= int(r["Start_Timestamp"]) < hi]
for r in tail:
    a = agg[r["Kernel_Name"][:110]]
    a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
with open(out + "/top_kernels.txt", "w") as g:
    g.write(f"# kernels of the last 20 frames of `bench.py --workload c5` (steady state, after MIOpen's find pass): {len(tail)} dispatches, "
            f"{tot / 20e3:.2f} ms of GPU time per frame, {(hi - lo) / 20e6:.2f} ms per frame start to start\n")
    g.write("# calls   total_us   share   kernel\n")
    for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        g.write(f"{n:6d} {us:10.1f} {us / tot:7.3f}   {k}\n")
print(open(out + "/top_kernels.txt").read())
PY
cp $(ls "$OUT"/trace/*/*_kernel_stats.csv | head -1) "$OUT/kernel_stats_bench_c5.csv" 2>/dev/null
