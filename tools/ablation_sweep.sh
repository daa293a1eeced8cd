#!/bin/bash
# Section costs of the fused kernels by ablation (timing only): every variant of
# tools/variants/abl_*.py x {verified trip table, 0, 1, 2, 3 trips on every curved surface}.
# Build first:  for v in ...; do tools/build_variant.sh <tag> --py tools/variants/abl_<...>.py; done
OUT=${1:-gpurun_out/r03/ablation.log}
mkdir -p "$(dirname "$OUT")"; : > "$OUT"
for lib in sdirt_amd/libsdirt_dp.so build/libsdirt_dp_nosplat.so build/libsdirt_dp_nosplat_norefract.so \
           build/libsdirt_dp_nosplat_notail.so build/libsdirt_dp_nosplat_norefract_notail.so; do
    for t in "" "--trips 0" "--trips 1" "--trips 2" "--trips 3"; do
        echo "### $lib $t" >> "$OUT"
        SDIRT_AMD_LIB=$lib python3 tools/kbench.py --reps 6 $t 2>/dev/null | grep lib= >> "$OUT"
    done
done
cat "$OUT"
