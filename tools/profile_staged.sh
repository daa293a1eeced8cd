#!/bin/bash
# Profiling recipe for `bench.py --workload staged` (run on the GPU box through gpurun):
#   tools/profile_staged.sh <commit> [round-dir, default r05] [ks, default 65]
# Pass 1: kernel trace + stats.  Then PMC counters, each group in its own run (never trace domains together
# with --pmc on this pool).  Output: gpurun_out/prof_<round>_staged/{trace,pmc_*}/, summary.json and the condensed
# pmc_staged_ks<ks>.json (bench.py reads it from profiles/<round>/) (per kernel: rocprofv3 average duration, HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE as
# MI355X_MICROARCH.md prescribes for gfx950, L2 hit rate).
set -u
COMMIT=${1:-unknown}; RND=${2:-r05}; KS=${3:-65}
OUT=gpurun_out/prof_${RND}_staged_ks${KS}
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python3 bench.py --workload staged --staged-ks $KS --staged-chain calls --steps 10 --warmup 2"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $CMD > "$OUT/trace.log" 2>&1
CMD="python3 bench.py --workload staged --staged-ks $KS --staged-chain calls --steps 3 --warmup 2"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $CMD > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $CMD > "$OUT/pmc_write.log" 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d "$OUT/pmc_tcc" -- $CMD > "$OUT/pmc_tcc.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_BUSY_CYCLES \
    --output-format csv -d "$OUT/pmc_a" -- $CMD > "$OUT/pmc_a.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_ATOMIC GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT/pmc_b" -- $CMD > "$OUT/pmc_b.log" 2>&1
python3 tools/summarize_prof.py "$OUT" --biggest-grid > "$OUT/summary.json"
python3 - "$OUT" "$COMMIT" "$KS" <<'PY'
import json, sys, time
sys.path.insert(0, ".")
import bench
out, commit = sys.argv[1:3]
s = json.load(open(f"{out}/summary.json"))
d = {"workload": "staged", "ks": int(sys.argv[3]), "n_points": bench.STAGED_N, "spp": bench.STAGED_SPP, "commit": commit, "source_hash": bench.source_hash(),
     "collected": time.strftime("%Y-%m-%d %H:%M UTC", time.gmtime()),
     "command": "rocprofv3 --kernel-trace --stats / --pmc <group> -- python3 bench.py --workload staged",
     "note": "HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024 per dispatch (gfx950: FETCH_SIZE counts half of a wide "
             "streaming read)", "kernels": {}}
for k, t in s["kernel_trace"].items():
    e = {"calls": t["calls"], "avg_us": t["avg_us"], "median_us": t["median_us"], "min_us": t["min_us"],
         "grid": t.get("Grid_Size_X"), "workgroup": t.get("Workgroup_Size_X"), "lds": t.get("LDS_Block_Size"),
         "vgpr": t.get("VGPR_Count")}
    p = s["pmc"].get(k, {})
    for c, v in p.items():
        e[c] = v["mean_last3"]
    if "FETCH_SIZE" in p and "WRITE_SIZE" in p:
        e["hbm_bytes_per_dispatch_mean_last3"] = (2 * p["FETCH_SIZE"]["mean_last3"] + p["WRITE_SIZE"]["mean_last3"]) * 1024
    if "TCC_HIT_sum" in p:
        e["l2_hit_rate"] = p["TCC_HIT_sum"]["mean_last3"] / max(1.0, p["TCC_HIT_sum"]["mean_last3"] + p["TCC_MISS_sum"]["mean_last3"])
    d["kernels"][k] = e
json.dump(d, open(f"{out}/pmc_staged_ks{sys.argv[3]}.json", "w"), indent=1)
print(json.dumps(d, indent=1))
PY
