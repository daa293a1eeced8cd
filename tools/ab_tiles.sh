#!/bin/bash
# A/B of k_psf_lr's LDS accumulator type (same box, interleaved): product library (double tiles where four workgroups
# per CU still fit) vs build/libsdirt_dp_floattiles.so (float tiles everywhere) [vs build/libsdirt_dp_wg1024.so].
#   tools/build_variant.sh floattiles --py tools/variants/psf_float_tiles.py
#   tools/build_variant.sh wg1024 --py tools/variants/psf_wg1024_wide.py
cd "$(dirname "$0")/.."
shapes=("--n 8192 --spp 8192 --ks 21" "--n 24576 --spp 4096 --ks 21" "--n 64 --spp 20000 --ks 21" "--n 16384 --spp 4096 --ks 45" "--n 16384 --spp 4096 --ks 65")
for round in 1 2 3; do
  for sh in "${shapes[@]}"; do
    for lib in "" build/libsdirt_dp_floattiles.so build/libsdirt_dp_wg1024.so; do
      [ -n "$lib" ] && [ ! -f "$lib" ] && continue
      SDIRT_AMD_LIB=$lib python tools/kbench.py --reps 8 $sh 2>&1 | grep "^lib="
    done
  done
done
