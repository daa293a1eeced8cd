#!/bin/bash
# PMC profile of the two hot kernels under tools/kbench.py (fixed trip tables, config-2 volume).
# Usage (on the GPU box, through gpurun):  tools/profile_kbench.sh <tag> [lib.so] [kbench args...]
#   -> gpurun_out/prof_<tag>/{trace,pmc_*}/...   then: python tools/summarize_prof.py gpurun_out/prof_<tag>
# One counter group per run; never trace domains together with --pmc on this pool.
set -u
TAG=$1; LIB=${2:-}; shift; shift || true
OUT=gpurun_out/prof_${TAG}
mkdir -p "$OUT"
export TMPDIR=/tmp
[ -n "$LIB" ] && export SDIRT_AMD_LIB=$LIB
CMD="python3 tools/kbench.py --reps 3 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $CMD > "$OUT/trace.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES \
    --output-format csv -d "$OUT/pmc_a" -- $CMD > "$OUT/pmc_a.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_IFETCH SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS \
    --output-format csv -d "$OUT/pmc_b" -- $CMD > "$OUT/pmc_b.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS_ATOMIC SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
    --output-format csv -d "$OUT/pmc_c" -- $CMD > "$OUT/pmc_c.log" 2>&1
rocprofv3 --pmc SQ_INST_LEVEL_SMEM SQ_IFETCH_LEVEL SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT \
    --output-format csv -d "$OUT/pmc_d" -- $CMD > "$OUT/pmc_d.log" 2>&1
python3 tools/summarize_prof.py "$OUT" > "$OUT/summary.json"
