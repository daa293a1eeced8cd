#!/bin/bash
# Profiling recipe for the bench workload (run on the GPU box through gpurun).
# Pass 1: kernel trace + stats.  Passes 2-4: PMC counters, each in its own run
# (no trace domains together with --pmc on this pool).
# Usage: tools/profile_r01.sh <tag>     -> gpurun_out/prof_<tag>/...
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_${TAG}
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $CMD > "$OUT/trace.log" 2>&1
CMD="python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY \
    --output-format csv -d "$OUT/pmc_sq1" -- $CMD > "$OUT/pmc_sq1.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT/pmc_sq2" -- $CMD > "$OUT/pmc_sq2.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $CMD > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $CMD > "$OUT/pmc_write.log" 2>&1
find "$OUT" -name "*.csv" | head -40
