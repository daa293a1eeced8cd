#!/bin/bash
# Kernel trace + HBM counters of the f1 workload (per-pixel DP-PSF convolution, bench.py --workload f1):
#   tools/profile_render.sh <commit> [round-dir, default r03]   -> gpurun_out/prof_<round>_f1/summary_render.json
set -u
COMMIT=${1:-unknown}; RND=${2:-r06}
OUT=gpurun_out/prof_${RND}_f1
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python3 bench.py --workload f1 --steps 50 --warmup 5 --sustain-seconds 0"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $CMD > "$OUT/trace.log" 2>&1
CMD="python3 bench.py --workload f1 --steps 5 --warmup 2 --sustain-seconds 0"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $CMD > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $CMD > "$OUT/pmc_write.log" 2>&1
# what the kernel does with its time beside the loads (round 6: why 0.70 of 8 TB/s when its read pattern alone reaches 0.87)
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES \
    --output-format csv -d "$OUT/pmc_a" -- $CMD > "$OUT/pmc_a.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS \
    --output-format csv -d "$OUT/pmc_b" -- $CMD > "$OUT/pmc_b.log" 2>&1
# the kernel's READ PATTERN alone (tools/bw_pixel.hip: one wave per pixel, 14 dword loads per lane one pixel ahead, no
# arithmetic, no LDS) under the same counters: what separates the product from it
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/bw_pixel.hip -o /tmp/bw_pixel > "$OUT/bw_pixel_build.log" 2>&1
/tmp/bw_pixel > "$OUT/bw_pixel.txt" 2>&1
rocprofv3 --kernel-trace --output-format csv -d "$OUT/probe_trace" -- /tmp/bw_pixel > "$OUT/probe_trace.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES \
    --output-format csv -d "$OUT/probe_pmc_a" -- /tmp/bw_pixel > "$OUT/probe_pmc_a.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS \
    --output-format csv -d "$OUT/probe_pmc_b" -- /tmp/bw_pixel > "$OUT/probe_pmc_b.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
res = {}
for f in glob.glob(out + "/probe_pmc_*/**/*_counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        for c, v in cs.items():
            res.setdefault(k, {})[c] = v[-1]
dur = collections.defaultdict(list)
for f in glob.glob(out + "/probe_trace/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in dur.items():
    res.setdefault(k, {})["min_us"] = min(v)
json.dump(res, open(out + "/probe_counters.json", "w"), indent=1)
PY
python3 tools/summarize_prof.py "$OUT" k_local_psf_render > "$OUT/summary.json"
python3 - "$OUT" "$COMMIT" <<'PY'
import json, sys, time
sys.path.insert(0, ".")
import bench
out, commit = sys.argv[1:3]
s = json.load(open(f"{out}/summary.json"))
name = next(k for k in s["kernel_trace"] if "render" in k)
t, p = s["kernel_trace"][name], s["pmc"][name]
alg = 512 * 768 * 2 * 441 * 4 + 3 * 512 * 768 * 4 * 3
hbm = (2 * p["FETCH_SIZE"]["last"] + p["WRITE_SIZE"]["last"]) * 1024      # KB units, gfx950 reports half of the reads
d = {"workload": "f1", "commit": commit, "source_hash": bench.source_hash(),
     "collected": time.strftime("%Y-%m-%d %H:%M UTC", time.gmtime()), "kernel": name,
     "kernel_trace_avg_us": t["avg_us"], "kernel_trace_median_us": t["median_us"], "calls": t["calls"],
     "algorithmic_bytes": alg, "hbm_bytes_counted": hbm, "traffic_over_algorithmic": hbm / alg,
     "achieved_GBs_avg": alg / (t["avg_us"] * 1e-6) / 1e9, "frac_of_8TBs": alg / (t["avg_us"] * 1e-6) / 8e12,
     "counters_last_dispatch": {k: v["last"] for k, v in p.items() if k not in ("FETCH_SIZE", "WRITE_SIZE")}}
json.dump(d, open(f"{out}/summary_render.json", "w"), indent=1)
print(json.dumps(d))
PY
