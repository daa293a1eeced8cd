#!/bin/bash
# Kernel A/B builds -- the product sources carry no experiment switches; a variant is the product
# tree (or an older revision of it) plus a patch, built in a scratch directory:
#
#   tools/build_variant.sh <tag> [--rev <git-rev>] [--patch <file.patch>]... [--py <edit.py>]... [-DFLAG ...]
#     --py: a Python script run as `python edit.py <variant-tree>/sdirt_amd/csrc` that rewrites sources there
#           (tools/variants/*.py: the ablations and form experiments behind profiles/r03/k_psf_lr_sites.txt)
#     -> build/libsdirt_dp_<tag>.so     (use: SDIRT_AMD_LIB=build/libsdirt_dp_<tag>.so python tools/kbench.py)
#
# The variant goes through the same Makefile as the product (same flags, same prefetch-hazard check
# on the ISA of that very compile); extra -D / -f flags are appended to CXXFLAGS.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
TAG=$1; shift
REV=""; PATCHES=(); EDITS=(); EXTRA=()
while [ $# -gt 0 ]; do
    case "$1" in
        --rev) REV=$2; shift 2 ;;
        --patch) PATCHES+=("$(realpath "$2")"); shift 2 ;;
        --py) EDITS+=("$(realpath "$2")"); shift 2 ;;
        *) EXTRA+=("$1"); shift ;;
    esac
done
W="$ROOT/build/variant_$TAG"
rm -rf "$W"; mkdir -p "$W/sdirt_amd" "$W/tools"
if [ -n "$REV" ]; then
    git -C "$ROOT" archive "$REV" sdirt_amd/csrc include | tar -x -C "$W"
else
    cp -r "$ROOT/sdirt_amd/csrc" "$W/sdirt_amd/csrc"; cp -r "$ROOT/include" "$W/include"
    rm -rf "$W/sdirt_amd/csrc/obj"
fi
cp "$ROOT/tools/check_prefetch_hazard.py" "$W/tools/"
cp "$ROOT/sdirt_amd/csrc/Makefile" "$W/sdirt_amd/csrc/Makefile.head"
for p in "${PATCHES[@]}"; do patch -d "$W" -p1 < "$p"; done
for e in "${EDITS[@]}"; do python3 "$e" "$W/sdirt_amd/csrc"; done
cd "$W/sdirt_amd/csrc"
if [ -f sdirt_dp.hip ]; then      # round-2 layout: one translation unit, its own Makefile
    FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize"
    for t in sdirt_dp sdirt_mlp sdirt_dfdp; do
        /opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS "${EXTRA[@]}" -c $t.hip -o $t.o &
    done; wait
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared sdirt_dp.o sdirt_mlp.o sdirt_dfdp.o -o "$ROOT/build/libsdirt_dp_$TAG.so"
else
    make -s -j5 OUT="$ROOT/build/libsdirt_dp_$TAG.so" \
        CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-parameter ${EXTRA[*]}"
fi
echo "built build/libsdirt_dp_$TAG.so"
