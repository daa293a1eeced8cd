#!/bin/bash
# Same-box A/B of the fused PSF network (tools/mlp_bench.py: 2 x 393216 rows, median of 12 calls, output checksums):
#   tools/ab_mlp.sh <tag> [<tag> ...]     -- the product library against build/libsdirt_dp_<tag>.so, three interleaved rounds
for r in 1 2 3; do
    echo "round $r"
    echo -n "  product: "; python tools/mlp_bench.py 2>/dev/null | tail -1
    for t in "$@"; do
        echo -n "  $t: "; SDIRT_AMD_LIB=build/libsdirt_dp_$t.so python tools/mlp_bench.py 2>/dev/null | tail -1
    done
done
