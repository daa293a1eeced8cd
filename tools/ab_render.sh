#!/bin/bash
# A/B of the per-pixel renderers (same box, interleaved): product library vs build/libsdirt_dp_<tag>.so ...
#   tools/ab_render.sh renderlb8 [more tags]
cd "$(dirname "$0")/.."
for round in 1 2 3; do
  for tag in "" "$@"; do
    lib=""; [ -n "$tag" ] && lib=build/libsdirt_dp_$tag.so
    echo "== round $round lib=${tag:-product}"
    SDIRT_AMD_LIB=$lib python bench.py --workload f1 --steps 50 --sustain-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('f1 kernel ms', round(d['kernels_ms']['local_psf_render'],4), 'frac', round(d['roofline']['frac'],3))"
    SDIRT_AMD_LIB=$lib python tools/psfnet_render_bench.py 2>/dev/null | tail -1
  done
done
