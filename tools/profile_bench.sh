#!/bin/bash
# Profiling recipe for the bench workloads (run on the GPU box through gpurun):
#   tools/profile_bench.sh <workload: c2|c3|c4> <commit> [round-dir, default r06]
# Pass 1: kernel trace + stats of `python3 bench.py` (the same command the driver runs, minus the
# CPU baseline).  Passes 2-4: PMC counters, each group in its own run (never trace domains
# together with --pmc on this pool).  Output: gpurun_out/prof_<round>_<workload>/ and the condensed
# gpurun_out/prof_<round>_<workload>/pmc_<workload>.json that bench.py reads from profiles/<round>/.
set -u
WL=${1:-c2}; COMMIT=${2:-unknown}; RND=${3:-r06}
OUT=gpurun_out/prof_${RND}_${WL}
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-also --no-two-streams --sustain-seconds 0 --workload $WL"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $CMD > "$OUT/trace.log" 2>&1
CMD="python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-also --no-two-streams --sustain-seconds 0 --workload $WL"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES \
    --output-format csv -d "$OUT/pmc_a" -- $CMD > "$OUT/pmc_a.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE \
    --output-format csv -d "$OUT/pmc_b" -- $CMD > "$OUT/pmc_b.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $CMD > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $CMD > "$OUT/pmc_write.log" 2>&1
python3 tools/summarize_prof.py "$OUT" > "$OUT/summary.json"
python3 - "$OUT" "$WL" "$COMMIT" <<'PY'
import json, sys, time
sys.path.insert(0, ".")
import bench
out, wl, commit = sys.argv[1:4]
s = json.load(open(f"{out}/summary.json"))
p, t = s["pmc"]["k_psf_lr"], s["kernel_trace"]["k_psf_lr"]
v = lambda k: p[k]["last"]
d = {"workload": wl, "commit": commit, "source_hash": bench.source_hash(), "collected": time.strftime("%Y-%m-%d %H:%M UTC", time.gmtime()),
     "command": f"rocprofv3 --pmc <group> -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --workload {wl}",
     "kernel": "k_psf_lr<R, small-r, Lean, CENTER> (last dispatch of each pass)",
     "kernel_trace_median_us": t["median_us"], "kernel_trace_avg_us": t["avg_us"], "kernel_trace_calls": t["calls"],
     # the first two launches of a process run the cold 10-trip table and the table it reveals (trip-table discovery):
     # the steady-state average leaves them out
     "kernel_trace_avg_steady_us": sum(t["per_dispatch_us"][2:]) / max(1, len(t["per_dispatch_us"][2:])),
     # HBM traffic as MI355X_MICROARCH.md prescribes: FETCH_SIZE / WRITE_SIZE are in KB, separate
     # passes, gfx950 reports half of the read bytes
     "FETCH_SIZE_KB": v("FETCH_SIZE"), "WRITE_SIZE_KB": v("WRITE_SIZE"),
     "k_psf_lr_hbm_bytes_per_launch": (2 * v("FETCH_SIZE") + v("WRITE_SIZE")) * 1024,
     "k_psf_lr_valu_wave_instructions_per_launch": v("SQ_INSTS_VALU"),
     "salu_wave_instructions_per_launch": v("SQ_INSTS_SALU"), "smem_instructions_per_launch": v("SQ_INSTS_SMEM"),
     "trans_f32_instructions_per_launch": v("SQ_INSTS_VALU_TRANS_F32"),
     # live lanes per vector instruction: thread-cycles / (64 * wave-cycles spent in VALU instructions)
     "valu_lane_utilisation": v("SQ_THREAD_CYCLES_VALU") / (64 * v("SQ_ACTIVE_INST_VALU")),
     "wave_cycles": {k: v(k) for k in ("SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY")},
     "shader_clock_ghz": v("GRBM_GUI_ACTIVE") / 8 / (t["median_us"] * 1e-6) / 1e9}
json.dump(d, open(f"{out}/pmc_{wl}.json", "w"), indent=1)
print(json.dumps(d))
PY
